"""Output arena: where the op's y / dx / z live (C ABI: `cnsn_arena_*`, include/cnsn_hip.h).

The reference's op returns new tensors (models/cnsn.py:29,150) and torch's caching allocator decides where they lie.  On
MI355X that decides 7-10 % of the single-touch launches' time: their plane-strided writes run at the copy rate into one
large `hipMalloc` block in five and 10-20 % below it into the rest (profiles/r04_memory_map.md).  The library can create
blocks it chooses among — address ranges mapped from physical allocations of its own, a NEW block of 384 MiB or more the
fastest of eight candidates created together and timed with a plane-strided fill (~1 ms each, `set_tries` /
`CNSN_ARENA_TRIES`; smaller blocks are never timed — the Infinity Cache absorbs a write of that size).

Since round 6 those blocks belong to TORCH's allocator: the arena is a `torch.cuda.MemPool` (one per device, `use_on_oom`)
whose segments come from `cnsn_arena_map` / `cnsn_arena_unmap`, and the op's outputs of at least `min_bytes` (default
32 MiB; `CNSN_ARENA_MIN_MB`) are ordinary tensors allocated while that pool is active for the calling thread.  So:
  * `torch.cuda.memory_allocated` / `memory_reserved` / `memory_snapshot` count them; `Tensor.record_stream` orders their
    re-use across streams like any tensor's (round 5's `at::from_blob` tensors were invisible to both);
  * the pool's free blocks are split and re-used by the caching allocator (a last, smaller batch does not pin a second
    full-size set), released by its out-of-memory path, and LENT to any other allocation of the process that would
    otherwise fail (`use_on_oom`): a model that fits with the reference's plain allocation fits with the arena on;
  * `trim()` releases the pool's free blocks (`torch.cuda.empty_cache()` leaves user pools alone);
  * on by default (`CNSN_ARENA=0` or `disable()` switch it off); outputs under graph capture and small outputs come from
    torch's default pool as before; if the driver cannot map memory the allocation falls back to it (`stats()['broken']`).
`prospect()` and the `cnsn_arena_alloc` family remain the C ABI's own caching layer (for callers without torch); the Python
layer no longer uses it for the op's outputs.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _ffi

__all__ = ["enable", "disable", "enabled", "min_bytes", "stats", "trim", "empty_like", "set_chunk_mb", "prospect", "block_gbps",
           "set_tries", "set_limit_mb", "pool_id"]


def _glue():
    g = _ffi.glue()
    if g is None:
        raise _ffi.CnsnError("the output arena hands out tensors through the C++ glue (cnsn_glue.so), which is not built / "
                             "was switched off with CNSN_NO_GLUE=1")
    return g


def enable(min_mb: float = 32.0) -> int:
    """outputs of at least `min_mb` MiB come from the arena; returns the previous threshold in bytes (-1: was off)"""
    return int(_glue().arena_config(int(min_mb * (1 << 20))))


def disable() -> int:
    g = _ffi.glue()
    return int(g.arena_config(-1)) if g is not None else -1


def min_bytes() -> int:
    """threshold in force in bytes, -1 when the arena is off (or the glue is not built)"""
    g = _ffi.glue()
    return int(g.arena_min_bytes()) if g is not None else -1


def enabled() -> bool:
    return min_bytes() >= 0


def out_like(x: torch.Tensor) -> torch.Tensor:
    """the allocation the op makes for an output shaped like the dense tensor `x` (functional.py's ctypes path)"""
    g = _ffi.glue()
    return g.out_like(x) if g is not None else torch.empty_like(x)


def empty_like(x: torch.Tensor) -> torch.Tensor:
    """a contiguous tensor of x's shape and type from the arena's pool, whatever its size"""
    return _glue().arena_empty_like(x)


def pool_id(device=None):
    """id of the `torch.cuda.MemPool` behind the arena on `device` — `torch.cuda.memory_snapshot(pool_id())` lists its
    segments; (0, 0) when the pool could not be created"""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    return tuple(int(v) for v in _glue().arena_pool_id(int(dev)))


def stats(device=None) -> dict:
    """the library's counters for `device` (blocks it created and still holds for torch's pool or its own cache)"""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    st = _ffi.ArenaStats()
    st.struct_bytes = C.sizeof(_ffi.ArenaStats)
    _ffi.check(_ffi.lib().cnsn_arena_stats(int(dev), C.byref(st)), "cnsn_arena_stats")
    return {k: int(getattr(st, k)) for k, _ in _ffi.ArenaStats._fields_ if k != "struct_bytes"}


def trim(device=None) -> int:
    """release every free block — of the torch pool (`emptyCache(pool)`) and of the C ABI's own cache — on `device` (all
    devices when None); returns the bytes of physical memory given back to the driver"""
    devs = range(torch.cuda.device_count()) if device is None else [torch.device(device).index]
    before = sum(stats(d)["mapped_bytes"] for d in devs)
    g = _ffi.glue()
    if g is not None:
        g.arena_trim(-1 if device is None else int(torch.device(device).index))
    _ffi.lib().cnsn_arena_trim(-1 if device is None else int(torch.device(device).index))
    return before - sum(stats(d)["mapped_bytes"] for d in devs)


def set_limit_mb(mb: float, device=None) -> None:
    """cap on what the C ABI's own caching layer (`cnsn_arena_alloc`) holds on `device`; 0: the default (CNSN_ARENA_MAX_MB, else
    half of the device memory).  The torch pool is bounded by torch's allocator, not by this."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    _ffi.check(_ffi.lib().cnsn_arena_set_limit(int(dev), int(mb * (1 << 20))), "cnsn_arena_set_limit")


def prospect(like, keep: int = 4, candidates: int = 12) -> dict:
    """Look for fast memory, explicitly and bounded (`cnsn_arena_prospect`, the C ABI's own cache): create `candidates` blocks
    of `like`'s size (a tensor, or a byte count), time a plane-strided fill into each, keep the `keep` fastest on the free list
    of `cnsn_arena_alloc` and give the rest back.  Transient memory: candidates x size, never more than half of what is free.
    A measurement aid since round 6 (the op's outputs come from the torch pool, whose new blocks are the best of
    `set_tries()` candidates each)."""
    if isinstance(like, torch.Tensor):
        nbytes, dev = like.numel() * like.element_size(), like.device
    else:
        nbytes, dev = int(like), torch.device("cuda", torch.cuda.current_device())
    rates = (C.c_float * int(candidates))()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    kept = _ffi.lib().cnsn_arena_prospect(int(dev.index), nbytes, int(keep), int(candidates), stream, rates)
    if kept < 0:
        _ffi.check(kept, "cnsn_arena_prospect")
    got = sorted((round(float(v), 1) for v in rates if v > 0), reverse=True)
    return {"bytes_per_block": nbytes, "candidates": len(got), "kept": int(kept), "GBps_fill": got,
            "GBps_kept_min": got[kept - 1] if 0 < kept <= len(got) else None,
            "GBps_median": got[len(got) // 2] if got else None}


def block_gbps(t: torch.Tensor) -> float:
    """the write rate measured for the arena block under `t` (0.0: never measured / not an arena tensor)"""
    v = C.c_float(0.0)
    return float(v.value) if _ffi.lib().cnsn_arena_block_gbps(C.c_void_p(t.data_ptr()), C.byref(v)) == 0 else 0.0


def set_tries(tries: int) -> int:
    """candidates the arena creates and times per NEW block of 384 MiB or more, keeping the fastest (default 8,
    `CNSN_ARENA_TRIES`; 1: none; smaller blocks are never timed); returns the previous value"""
    return int(_ffi.lib().cnsn_arena_set_tries(int(tries)))


def set_chunk_mb(mb: float) -> None:
    """size of the physical allocations NEW blocks are mapped from (measurement knob; 0: default)"""
    _ffi.check(_ffi.lib().cnsn_arena_set_chunk_bytes(int(mb * (1 << 20))), "cnsn_arena_set_chunk_bytes")
