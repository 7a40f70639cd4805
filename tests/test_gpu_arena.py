"""cnsn_amd.arena — the op's outputs live in address ranges mapped from small physical allocations (C ABI `cnsn_arena_*`,
include/cnsn_hip.h; why: profiles/r04_memory_map.md, profiles/r05_arena.md).  The reference's op returns new tensors
(models/cnsn.py:29,150): WHERE they lie must be invisible in the results and in the tensors' behaviour."""
import ctypes as C
import gc
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from cnsn_amd import _ffi, arena  # noqa: E402
from cnsn_amd import functional as F_  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402

DEV = torch.device("cuda:0")


@pytest.fixture(autouse=True)
def default_arena():
    was = arena.min_bytes()
    yield
    _ffi.glue().arena_config(was)
    arena.set_chunk_mb(0)
    arena.set_tries(0)
    gc.collect()
    if os.environ.get("CNSN_TEST_NO_TRIM") != "1":
        arena.trim()


def _owned(t):
    return cnsn_amd.lib().cnsn_arena_owns(C.c_void_p(t.data_ptr())) == 1


def test_c_abi_alloc_free_reuse_and_trim():
    lib = cnsn_amd.lib()
    arena.trim()
    s0 = arena.stats(DEV)
    stream = C.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
    p = lib.cnsn_arena_alloc(0, 100 << 20, stream)
    assert p and p % 16 == 0 and lib.cnsn_arena_owns(C.c_void_p(p)) == 1 and lib.cnsn_arena_owns(C.c_void_p(p + (100 << 20) - 1)) == 1
    s1 = arena.stats(DEV)
    assert s1["misses"] == s0["misses"] + 1 and s1["blocks_in_use"] == s0["blocks_in_use"] + 1
    assert s1["chunk_bytes"] == 56 << 20 and s1["in_use_bytes"] - s0["in_use_bytes"] == 2 * (56 << 20)      # whole chunks
    assert lib.cnsn_arena_free(C.c_void_p(p)) == 0
    assert lib.cnsn_arena_free(C.c_void_p(p)) < 0                               # not handed out: refused, nothing corrupted
    assert lib.cnsn_arena_free(C.c_void_p(12345)) < 0
    q = lib.cnsn_arena_alloc(0, 90 << 20, stream)                               # same number of chunks: the block comes back
    assert q == p and arena.stats(DEV)["hits"] == s1["hits"] + 1
    assert lib.cnsn_arena_alloc(0, 0, stream) is None
    failed = arena.stats(DEV)["failed"]
    assert lib.cnsn_arena_alloc(0, 400 << 30, stream) is None                   # more than the part has: NULL, the caller falls back
    assert arena.stats(DEV)["failed"] == failed + 1
    again = lib.cnsn_arena_alloc(0, 10 << 20, stream)                           # ... and the arena goes on serving
    assert again and lib.cnsn_arena_free(C.c_void_p(again)) == 0
    lib.cnsn_arena_free(C.c_void_p(q))
    freed = arena.trim(DEV)
    assert freed >= 2 * (56 << 20) and arena.stats(DEV)["mapped_bytes"] == arena.stats(DEV)["in_use_bytes"]
    assert lib.cnsn_arena_owns(C.c_void_p(p)) == 0
    r = lib.cnsn_arena_alloc(0, 100 << 20, stream)                              # a NEW address range: an unmapped one is never
    assert r and r != p                                                         # mapped again (stale translations, see below)
    lib.cnsn_arena_free(C.c_void_p(r))


def test_trimmed_address_ranges_are_never_mapped_again():
    """ROCm 7.2: a range handed back with hipMemUnmap + hipMemAddressFree, re-reserved and mapped to OTHER physical memory was
    accessed through stale translations — y and dx of a launch ended up in each other's old pages (seen with 2 MiB chunks
    right after a trim: profiles/r05_arena.md).  The arena therefore keeps every range it ever mapped reserved; results after
    any number of trim / re-create rounds are the bits the plain allocator gives."""
    shape = (40, 16, 56, 56)
    torch.manual_seed(5)
    np.random.seed(5)
    x = torch.randn(shape, device=DEV) * 1.3 + 0.2
    gy = torch.randn(shape, device=DEV)
    d = cnsn_amd.draw_cn(shape, "both", 1)

    def run():
        mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm("both", 1), fill_sn(cnsn_amd.SelfNorm(16), 7, torch.float32)).to(DEV).train()
        mod.crossnorm.active = True
        mod.crossnorm.next_draws = d
        xg = x.clone().requires_grad_()
        y = mod(xg)
        y.backward(gy)
        torch.cuda.synchronize()
        return y.detach().clone(), xg.grad.clone(), y.data_ptr()

    arena.disable()
    y0, g0, _ = run()
    seen = set()
    for rnd in range(6):
        arena.set_chunk_mb(2 if rnd % 2 else 0)
        arena.enable(min_mb=1)
        for _ in range(2):
            y, g, ptr = run()
            assert torch.equal(y, y0) and torch.equal(g, g0), rnd
        assert ptr not in seen
        seen.add(ptr)
        del y, g
        gc.collect()
        assert arena.trim(DEV) > 0


def test_a_new_large_block_is_the_fastest_of_its_candidates():
    """the arena's standing policy (cnsn_arena_set_tries, default 8) for blocks of 384 MiB or more: candidates created together,
    each timed with the plane-strided fill, the fastest kept, the others' memory back to the driver at once.  Smaller blocks are never timed: the Infinity Cache absorbs a write of that size."""
    arena.trim()
    assert arena.set_tries(0) in (1, 2, 3, 4, 6, 8, 32) and arena.set_tries(0) == 8      # (0: back to the default, which is 8)
    arena.set_tries(4)
    small = torch.randn(64, 64, 56, 56, device=DEV)                            # 51 MB
    s0 = arena.stats(DEV)
    a = arena.empty_like(small)
    s1 = arena.stats(DEV)
    assert s1["probed"] == s0["probed"] and s1["blocks"] == s0["blocks"] + 1 and arena.block_gbps(a) == 0.0
    big = torch.empty(128, 256, 56, 56, device=DEV)                            # 392 MiB: 7 chunks of 56 MiB
    b = arena.empty_like(big)
    s2 = arena.stats(DEV)
    assert s2["probed"] == s1["probed"] + 4 and s2["blocks"] == s1["blocks"] + 1 and s2["misses"] == s1["misses"] + 1
    assert s2["mapped_bytes"] - s1["mapped_bytes"] == 392 << 20                # the three losers are gone
    assert arena.block_gbps(b) > 100.0
    del b
    gc.collect()
    c = arena.empty_like(big)                                                  # steady state: a list pop, nothing timed
    assert arena.stats(DEV)["probed"] == s2["probed"] and arena.stats(DEV)["hits"] == s2["hits"] + 1
    arena.set_tries(1)
    d = arena.empty_like(big)
    assert arena.stats(DEV)["probed"] == s2["probed"] and arena.block_gbps(d) == 0.0
    del a, c, d
    gc.collect()
    arena.trim()
    arena.set_tries(0)


def test_prospect_keeps_the_fastest_blocks_on_the_free_list():
    arena.trim()
    x = torch.randn(64, 64, 56, 56, device=DEV)                                # 51 MB
    s0 = arena.stats(DEV)
    rep = arena.prospect(x, keep=2, candidates=6)
    s1 = arena.stats(DEV)
    assert rep["candidates"] == 6 and rep["kept"] == 2 and len(rep["GBps_fill"]) == 6
    assert rep["GBps_fill"] == sorted(rep["GBps_fill"], reverse=True) and rep["GBps_fill"][-1] > 100.0
    assert s1["blocks"] == s0["blocks"] + 2 and s1["blocks_in_use"] == s0["blocks_in_use"]      # the other four went back
    a, b = arena.empty_like(x), arena.empty_like(x)                            # the kept blocks, fastest first
    assert arena.stats(DEV)["misses"] == s1["misses"]
    ga, gb = arena.block_gbps(a), arena.block_gbps(b)
    assert ga >= gb > 0 and round(ga, 1) == rep["GBps_fill"][0] and round(gb, 1) == rep["GBps_fill"][1]
    arena.set_tries(1)
    c = arena.empty_like(x)                                                    # a third one: created with one try, never measured
    assert arena.block_gbps(c) == 0.0
    del a
    gc.collect()
    assert arena.block_gbps(arena.empty_like(x)) == ga                         # a measured free block goes first


def test_outputs_above_the_threshold_come_from_the_arena_and_go_back_to_it():
    arena.enable(min_mb=8)
    x = torch.randn(32, 32, 56, 56, device=DEV, requires_grad=True)            # 12.8 MB
    mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm("neither", 1), fill_sn(cnsn_amd.SelfNorm(32), 3, torch.float32)).to(DEV).train()
    mod.crossnorm.active = True
    y = mod(x)
    assert _owned(y) and y.is_contiguous() and y.shape == x.shape and y.dtype == x.dtype and y.device == x.device
    gx, = torch.autograd.grad(y, [x], torch.randn_like(y))
    assert _owned(gx)
    ptr = y.data_ptr()
    in_use = arena.stats(DEV)["blocks_in_use"]
    z = (y * 2).sum()                          # torch ops read arena tensors like any other
    assert torch.isfinite(z)
    v = y.view(32, -1)[3:5]                     # views keep the block alive
    del y
    assert arena.stats(DEV)["blocks_in_use"] == in_use
    del v, gx
    gc.collect()
    assert arena.stats(DEV)["blocks_in_use"] == in_use - 2
    with torch.no_grad():
        y2 = mod(x)                             # the freed block serves the next call of the same size
    assert y2.data_ptr() in (ptr, ) or _owned(y2)
    small = torch.randn(2, 32, 56, 56, device=DEV)
    with torch.no_grad():
        assert not _owned(mod(small))           # below the threshold: torch's allocator, as before
    arena.disable()
    with torch.no_grad():
        assert not _owned(mod(x))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("crop", ["neither", "both"])
def test_results_do_not_depend_on_the_arena(dtype, crop):
    """same launches, same bytes — wherever y / dx lie (arena of two chunk sizes vs torch's allocator), both glue paths"""
    shape = (40, 16, 56, 56)
    torch.manual_seed(5)
    np.random.seed(5)
    x = (torch.randn(shape, device=DEV) * 1.3 + 0.2).to(dtype)
    gy = torch.randn(shape, device=DEV).to(dtype)
    d = cnsn_amd.draw_cn(shape, crop, 1)

    def run(ctypes_path):
        sn = fill_sn(cnsn_amd.SelfNorm(shape[1]), 7, torch.float32).to(DEV).train()
        xg = x.clone().requires_grad_()
        if ctypes_path:
            kw, g, f = sn._fused_args()
            cfg = cnsn_amd.FusedConfig(cn_active=True, content_box=d.content_box, style_box=d.style_box, **kw)
            y = F_.FusedCNSN.apply(xg, cfg, d.perm, None, g.fc_weight, g.bn_weight, g.bn_bias, g.running_mean, g.running_var,
                                   *(None,) * 5, None, g.num_batches_tracked, None)
        else:
            mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1), sn).to(DEV).train()
            mod.crossnorm.active = True
            mod.crossnorm.next_draws = d
            y = mod(xg)
        y.backward(gy)
        torch.cuda.synchronize()
        return [y.detach().clone(), xg.grad.clone(), *(p.grad.clone() for p in sn.parameters()), sn.g_bn.running_var.clone()], _owned(y)

    arena.disable()
    base, owned = run(False)
    assert not owned
    for chunk in (0, 2):
        arena.set_chunk_mb(chunk)
        arena.enable(min_mb=1)
        for ctypes_path in (False, True):
            got, owned = run(ctypes_path)
            assert owned
            for i, (a, b) in enumerate(zip(base, got)):
                assert torch.equal(a, b), (dtype, crop, chunk, ctypes_path, i)


def test_fused_block_and_building_blocks_use_it_too():
    arena.enable(min_mb=1)
    x = torch.randn(16, 32, 56, 56, device=DEV, requires_grad=True)
    add = torch.randn_like(x)
    sn = fill_sn(cnsn_amd.SelfNorm(32), 2, torch.float32).to(DEV).train()
    kw, g, f = sn._fused_args()
    cfg = cnsn_amd.FusedConfig(add_mode="post", relu=True, **kw)
    y = F_.fused_cnsn(x, cfg, g=g, addend=add.requires_grad_())
    assert _owned(y)
    gx, ga = torch.autograd.grad(y, [x, add], torch.randn_like(y))
    assert _owned(gx) and _owned(ga)
    out = cnsn_amd.instance_norm_mix(x.detach(), torch.randn_like(x))
    assert _owned(out)


def test_a_block_last_used_on_another_stream_is_ordered_behind_it():
    """a block freed after work was queued on stream A and handed out on stream B: B's launch must not overtake A's reads"""
    arena.enable(min_mb=1)
    x = torch.randn(64, 64, 56, 56, device=DEV)
    sn = fill_sn(cnsn_amd.SelfNorm(64), 4, torch.float32).to(DEV).eval()
    a, b = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
    torch.cuda.synchronize()
    with torch.no_grad():
        want = sn(x).clone()
        torch.cuda.synchronize()
        for _ in range(10):
            with torch.cuda.stream(a):
                y = sn(x)                               # block handed out on A
                acc = y.clone()
                for _ in range(20):                     # a queue of reads of y on A
                    acc = torch.maximum(acc, y)
                ptr = y.data_ptr()
                del y                                   # freed while A still has work queued that reads it
            with torch.cuda.stream(b):
                t = _ffi.glue().arena_empty_like(x)     # the same block, now on B ...
                assert t.data_ptr() == ptr
                t.fill_(float("nan"))                   # ... overwritten at once
            torch.cuda.synchronize()
            assert torch.equal(acc, want)
            del t


def test_graph_capture_stays_on_torchs_allocator():
    arena.enable(min_mb=1)
    x = torch.randn(16, 32, 56, 56, device=DEV)
    sn = fill_sn(cnsn_amd.SelfNorm(32), 4, torch.float32).to(DEV).eval()
    with torch.no_grad():
        eager = sn(x)
        assert _owned(eager)
        s = torch.cuda.Stream(DEV)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            sn(x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = sn(x)
        assert not _owned(y)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, eager)
