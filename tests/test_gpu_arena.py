"""cnsn_amd.arena — the op's outputs live in address ranges mapped from small physical allocations (C ABI `cnsn_arena_*`,
include/cnsn_hip.h; why: profiles/r04_memory_map.md, profiles/r05_arena.md), handed out through a torch MemPool since round 6.  The reference's op returns new tensors
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
    """the C ABI's own caching layer (callers without a torch allocator to plug into)"""
    lib = cnsn_amd.lib()
    arena.trim()
    s0 = arena.stats(DEV)
    stream = C.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
    p = lib.cnsn_arena_alloc(0, 100 << 20, stream)
    assert p and p % 16 == 0 and lib.cnsn_arena_owns(C.c_void_p(p)) == 1 and lib.cnsn_arena_owns(C.c_void_p(p + (100 << 20) - 1)) == 1
    s1 = arena.stats(DEV)
    assert s1["misses"] == s0["misses"] + 1 and s1["blocks_in_use"] == s0["blocks_in_use"] + 1
    # one 56 MiB chunk + a 44 MiB tail chunk: what is held is the request rounded to the driver's granularity, not whole chunks
    assert s1["chunk_bytes"] == 56 << 20 and s1["in_use_bytes"] - s0["in_use_bytes"] == 100 << 20
    assert lib.cnsn_arena_record_stream(C.c_void_p(p), stream) == 0
    assert lib.cnsn_arena_free(C.c_void_p(p)) == 0
    assert lib.cnsn_arena_free(C.c_void_p(p)) < 0                               # not handed out: refused, nothing corrupted
    assert lib.cnsn_arena_record_stream(C.c_void_p(p), stream) < 0
    assert lib.cnsn_arena_free(C.c_void_p(12345)) < 0
    q = lib.cnsn_arena_alloc(0, 90 << 20, stream)                               # within an eighth of a free block: it comes back
    assert q == p and arena.stats(DEV)["hits"] == s1["hits"] + 1
    lib.cnsn_arena_free(C.c_void_p(q))
    small = lib.cnsn_arena_alloc(0, 40 << 20, stream)                           # the 100 MiB block would waste more than 1/8: new
    assert small and small != p and arena.stats(DEV)["misses"] == s1["misses"] + 1
    lib.cnsn_arena_free(C.c_void_p(small))
    assert lib.cnsn_arena_alloc(0, 0, stream) is None
    failed = arena.stats(DEV)["failed"]
    assert lib.cnsn_arena_alloc(0, 400 << 30, stream) is None                   # more than the part has: NULL, the caller falls back
    st = arena.stats(DEV)
    assert st["failed"] == failed + 1 and st["broken"] == 0                     # ... out of memory is not "does not work here"
    again = lib.cnsn_arena_alloc(0, 10 << 20, stream)                           # ... and the arena goes on serving
    assert again and lib.cnsn_arena_free(C.c_void_p(again)) == 0
    freed = arena.trim(DEV)
    assert freed >= (150 << 20) and arena.stats(DEV)["mapped_bytes"] == arena.stats(DEV)["in_use_bytes"]
    assert lib.cnsn_arena_owns(C.c_void_p(p)) == 0
    r = lib.cnsn_arena_alloc(0, 100 << 20, stream)                              # a NEW address range: an unmapped one is never
    assert r and r != p                                                         # mapped again (stale translations, see below)
    lib.cnsn_arena_free(C.c_void_p(r))


def test_c_abi_cache_is_capped_and_evicts_the_least_recently_used():
    """review of round 5: the cache had no cap, no eviction and exact-size lists only.  cnsn_arena_set_limit (default: half of
    the device memory, CNSN_ARENA_MAX_MB): free blocks go, least recently freed first, when a new block would exceed the cap;
    with everything in use the request is refused and the caller allocates as it always did."""
    lib = cnsn_amd.lib()
    arena.trim()
    stream = C.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
    base = arena.stats(DEV)["mapped_bytes"]
    arena.set_limit_mb((base >> 20) + 300)
    try:
        a = lib.cnsn_arena_alloc(0, 100 << 20, stream)
        b = lib.cnsn_arena_alloc(0, 60 << 20, stream)
        lib.cnsn_arena_free(C.c_void_p(a))                                      # a is the older free block
        lib.cnsn_arena_free(C.c_void_p(b))
        s0 = arena.stats(DEV)
        c = lib.cnsn_arena_alloc(0, 200 << 20, stream)                          # 160 + 200 > 300: a goes, b may stay
        s1 = arena.stats(DEV)
        assert c and s1["evicted"] == s0["evicted"] + 1 and lib.cnsn_arena_owns(C.c_void_p(a)) == 0
        assert lib.cnsn_arena_owns(C.c_void_p(b)) == 1 and s1["mapped_bytes"] - base == 260 << 20
        d = lib.cnsn_arena_alloc(0, 90 << 20, stream)                           # 260 + 90 > 300: b goes too
        assert d and arena.stats(DEV)["evicted"] == s0["evicted"] + 2
        failed = arena.stats(DEV)["failed"]
        assert lib.cnsn_arena_alloc(0, 64 << 20, stream) is None                # everything in use, the cap holds
        assert arena.stats(DEV)["failed"] == failed + 1 and arena.stats(DEV)["broken"] == 0
        lib.cnsn_arena_free(C.c_void_p(c))
        arena.set_limit_mb((base >> 20) + 100)                                  # a lower cap releases free blocks at once
        assert lib.cnsn_arena_owns(C.c_void_p(c)) == 0
        lib.cnsn_arena_free(C.c_void_p(d))
    finally:
        arena.set_limit_mb(0)
        arena.trim()


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
    assert s1["probed"] == s0["probed"] and s1["blocks"] == s0["blocks"] + 1 and arena.block_gbps(a) == 0.0 and _owned(a)
    big = torch.empty(128, 256, 56, 56, device=DEV)                            # 392 MiB
    b = arena.empty_like(big)
    s2 = arena.stats(DEV)
    assert s2["probed"] == s1["probed"] + 4 and s2["blocks"] == s1["blocks"] + 1 and s2["misses"] == s1["misses"] + 1
    assert s2["mapped_bytes"] - s1["mapped_bytes"] == 392 << 20                # the three losers are gone
    assert arena.block_gbps(b) > 100.0
    ptr = b.data_ptr()
    del b
    gc.collect()
    c = arena.empty_like(big)                                                  # steady state: torch's cache, nothing created or timed
    s3 = arena.stats(DEV)
    assert c.data_ptr() == ptr and s3["probed"] == s2["probed"] and s3["misses"] == s2["misses"]
    arena.set_tries(1)
    d = arena.empty_like(big)
    assert arena.stats(DEV)["probed"] == s2["probed"] and arena.block_gbps(d) == 0.0
    del a, c, d
    gc.collect()
    assert arena.trim() >= 2 * (392 << 20)
    arena.set_tries(0)


def test_prospect_keeps_the_fastest_blocks_on_the_free_list():
    """(the C ABI's own cache: cnsn_arena_prospect feeds cnsn_arena_alloc)"""
    lib = cnsn_amd.lib()
    arena.trim()
    x = torch.randn(64, 64, 56, 56, device=DEV)                                # 51 MB
    nbytes = x.numel() * 4
    stream = C.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
    s0 = arena.stats(DEV)
    rep = arena.prospect(x, keep=2, candidates=6)
    s1 = arena.stats(DEV)
    assert rep["candidates"] == 6 and rep["kept"] == 2 and len(rep["GBps_fill"]) == 6
    assert rep["GBps_fill"] == sorted(rep["GBps_fill"], reverse=True) and rep["GBps_fill"][-1] > 100.0
    assert s1["blocks"] == s0["blocks"] + 2 and s1["blocks_in_use"] == s0["blocks_in_use"]      # the other four went back

    def gbps(p):
        v = C.c_float(0.0)
        assert lib.cnsn_arena_block_gbps(C.c_void_p(p), C.byref(v)) == 0
        return float(v.value)

    a, b = lib.cnsn_arena_alloc(0, nbytes, stream), lib.cnsn_arena_alloc(0, nbytes, stream)    # the kept blocks, fastest first
    assert arena.stats(DEV)["misses"] == s1["misses"]
    ga, gb = gbps(a), gbps(b)
    assert ga >= gb > 0 and round(ga, 1) == rep["GBps_fill"][0] and round(gb, 1) == rep["GBps_fill"][1]
    c = lib.cnsn_arena_alloc(0, nbytes, stream)                                # a third one: below the timed size, never measured
    assert gbps(c) == 0.0
    lib.cnsn_arena_free(C.c_void_p(a))
    assert lib.cnsn_arena_alloc(0, nbytes, stream) == a                        # a measured free block goes first
    for p in (a, b, c):
        lib.cnsn_arena_free(C.c_void_p(p))


def test_outputs_above_the_threshold_come_from_the_arena_and_torch_sees_them():
    """review of round 5, "outputs that torch's allocator knows about" (models/cnsn.py:29,150 return ordinary tensors): the
    arena's blocks are segments of a torch MemPool — counted, cached, split and released by the caching allocator"""
    arena.enable(min_mb=8)
    x = torch.randn(32, 32, 56, 56, device=DEV, requires_grad=True)            # 12.8 MB
    mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm("neither", 1), fill_sn(cnsn_amd.SelfNorm(32), 3, torch.float32)).to(DEV).train()
    mod.crossnorm.active = True
    nbytes = x.numel() * 4
    m0 = torch.cuda.memory_allocated(DEV)
    y = mod(x)
    assert _owned(y) and y.is_contiguous() and y.shape == x.shape and y.dtype == x.dtype and y.device == x.device
    assert torch.cuda.memory_allocated(DEV) - m0 >= nbytes                     # torch.cuda.memory_allocated counts the output
    gx, = torch.autograd.grad(y, [x], torch.randn_like(y))
    assert _owned(gx)
    segs = torch.cuda.memory_snapshot(arena.pool_id(DEV))
    assert segs and all(cnsn_amd.lib().cnsn_arena_owns(C.c_void_p(s["address"])) == 1 for s in segs)
    ptr = y.data_ptr()
    z = (y * 2).sum()                          # torch ops read arena tensors like any other
    assert torch.isfinite(z)
    v = y.view(32, -1)[3:5]                     # views keep the block alive
    del y
    m1 = torch.cuda.memory_allocated(DEV)
    del v
    gc.collect()
    assert m1 - torch.cuda.memory_allocated(DEV) >= nbytes
    with torch.no_grad():
        y2 = mod(x)                             # the freed block serves the next call of the same size
    assert y2.data_ptr() == ptr
    y2.record_stream(torch.cuda.Stream(DEV))    # ... and record_stream means what it means for any tensor
    small = torch.randn(2, 32, 56, 56, device=DEV)
    with torch.no_grad():
        assert not _owned(mod(small))           # below the threshold: torch's default pool, as before
    arena.disable()
    with torch.no_grad():
        assert not _owned(mod(x))


def test_a_smaller_batch_reuses_the_blocks_of_the_full_one():
    """review of round 5: exact-size free lists made every distinct batch size (the last, partial batch of an epoch) pin a
    block set of its own.  The caching allocator splits the pool's free blocks instead."""
    arena.enable(min_mb=1)
    arena.trim()
    sn = fill_sn(cnsn_amd.SelfNorm(64), 4, torch.float32).to(DEV).eval()
    with torch.no_grad():
        for n in (64, 64):
            y = sn(torch.randn(n, 64, 56, 56, device=DEV))
            assert _owned(y)
            del y
        mapped = arena.stats(DEV)["mapped_bytes"]
        for n in (48, 33, 64, 17, 50):
            y = sn(torch.randn(n, 64, 56, 56, device=DEV))
            assert _owned(y)
            del y
        assert arena.stats(DEV)["mapped_bytes"] == mapped


def test_free_arena_blocks_do_not_cause_an_out_of_memory_error():
    """review of round 5: blocks the arena held were invisible to torch's free-cache-and-retry, so a model near the HBM limit
    that fits with the reference's plain allocation could fail with the arena on.  Now: an allocation that does not fit next to
    the pool's FREE blocks gets them (use_on_oom / the allocator's out-of-memory path)."""
    arena.enable(min_mb=1)
    arena.trim()
    gc.collect()
    torch.cuda.empty_cache()
    like = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
    a = arena.empty_like(like)                                                  # 1 GiB block of the arena, then free in its pool
    assert _owned(a)
    del a, like
    gc.collect()
    torch.cuda.empty_cache()
    free_b, _ = torch.cuda.mem_get_info(DEV)
    filler = torch.empty(free_b - (512 << 20), dtype=torch.uint8, device=DEV)   # the device is full but for 512 MiB
    try:
        t = torch.empty(900 << 20, dtype=torch.uint8, device=DEV)               # fits only with the arena's idle GiB
        t.fill_(3)
        torch.cuda.synchronize()
        assert int(t[-1]) == 3
        del t
    finally:
        del filler
        gc.collect()
        torch.cuda.empty_cache()
        arena.trim()


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


def test_a_block_used_on_another_stream_is_protected_by_record_stream():
    """review of round 5: `y.record_stream(s)` was a no-op on the arena's from_blob tensors (the caching allocator ignores
    pointers it did not allocate), so a block could be re-used while another stream still read it.  As pool blocks they follow
    the allocator's rules: a block freed while stream B's recorded work is pending is not handed out again before that work is
    done, and a request on another stream never gets a block of stream A."""
    arena.enable(min_mb=1)
    x = torch.randn(64, 64, 56, 56, device=DEV)
    sn = fill_sn(cnsn_amd.SelfNorm(64), 4, torch.float32).to(DEV).eval()
    b = torch.cuda.Stream(DEV)
    torch.cuda.synchronize()
    with torch.no_grad():
        want = sn(x).clone()
        torch.cuda.synchronize()
        for _ in range(10):
            y = sn(x)                                   # block handed out on the current stream
            assert _owned(y)
            ptr = y.data_ptr()
            b.wait_stream(torch.cuda.current_stream(DEV))
            with torch.cuda.stream(b):
                acc = y.clone()
                for _ in range(20):                     # a queue of reads of y on B
                    acc = torch.maximum(acc, y)
            y.record_stream(b)
            del y                                       # freed while B still has work queued that reads it
            t = arena.empty_like(x)                     # the next request on the first stream ...
            t.fill_(float("nan"))                       # ... overwrites its block at once
            assert _owned(t) and t.data_ptr() != ptr    # (not y's block: B's reads are still pending)
            with torch.cuda.stream(b):
                u = arena.empty_like(x)                 # a request on B never gets a block that belongs to another stream
                assert u.data_ptr() not in (ptr, t.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(acc, want)
            del t, u


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
