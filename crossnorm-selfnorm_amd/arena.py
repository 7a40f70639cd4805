"""Output arena: where the op's y / dx / z live (C ABI: `cnsn_arena_*`, include/cnsn_hip.h).

The reference's op returns new tensors (models/cnsn.py:29,150) and torch's caching allocator decides where they lie.  On
MI355X that decides 7-10 % of the single-touch launches' time: their plane-strided writes run at the copy rate into one
large `hipMalloc` block in five and 10-20 % below it into the rest (profiles/r04_memory_map.md).  Address ranges MAPPED from
physical allocations of the arena's own give the outputs a home that does not change from step to step: the op's outputs of at
least `min_bytes` (default 32 MiB; `CNSN_ARENA_MIN_MB`) are tensors over such ranges — `at::from_blob` views whose deleter
hands the block back to the arena's per-size free list.  A NEW block of 384 MiB or more is the fastest of eight candidates
created together and timed with a plane-strided fill (~1 ms each, `set_tries` / `CNSN_ARENA_TRIES`; where a block lies
physically decides how fast it is written, and about one in five lies well; smaller blocks are never timed — the Infinity
Cache absorbs a write of that size) — paid in the first steps of a job, for EVERY large output block of the job; in the
steady state of a training loop an allocation is a mutex and a list pop.

What a user may want to know:
  * on by default (`CNSN_ARENA=0` or `arena.disable()` switch it off); outputs under graph capture and small outputs come
    from torch's allocator as before; if the driver cannot map memory the call falls back silently (`stats()['failed']`);
  * WHERE a block lies physically decides how fast it is written, not what it is composed of (profiles/r05_arena.md):
    the arena gives the outputs a STABLE home, and `prospect()` is the explicit, bounded way to look for fast blocks;
  * blocks the arena holds are NOT visible to torch's allocator (`torch.cuda.memory_allocated` does not count them,
    `torch.cuda.empty_cache()` does not free them): `arena.stats()` / `arena.trim()` are the counterparts;
  * stream semantics are a caching allocator's: a block is re-used at once on the stream it was last used on, and behind an
    event on any other stream.  A tensor handed to ANOTHER stream and freed there needs the care `record_stream` asks for
    with torch's allocator (keep a reference until that stream is done).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _ffi

__all__ = ["enable", "disable", "enabled", "min_bytes", "stats", "trim", "empty_like", "set_chunk_mb", "prospect", "block_gbps", "set_tries"]


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
    """a contiguous tensor of x's shape and type over an arena block, whatever its size"""
    return _glue().arena_empty_like(x)


def stats(device=None) -> dict:
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    st = _ffi.ArenaStats()
    st.struct_bytes = C.sizeof(_ffi.ArenaStats)
    _ffi.check(_ffi.lib().cnsn_arena_stats(int(dev), C.byref(st)), "cnsn_arena_stats")
    return {k: int(getattr(st, k)) for k, _ in _ffi.ArenaStats._fields_ if k != "struct_bytes"}


def trim(device=None) -> int:
    """unmap and release every free block (all devices when `device` is None); returns the bytes released"""
    return int(_ffi.lib().cnsn_arena_trim(-1 if device is None else int(torch.device(device).index)))


def prospect(like, keep: int = 4, candidates: int = 12) -> dict:
    """Look for fast memory, explicitly and bounded (`cnsn_arena_prospect`): create `candidates` blocks of `like`'s size (a
    tensor, or a byte count), time a plane-strided fill into each, keep the `keep` fastest on the arena's free list — the
    next outputs of that size are written there — and give the rest back.  Transient memory: candidates x size, never more
    than half of what is free.  A job calls it once per large output size after building its model, or not at all."""
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
    """size of the physical allocations NEW blocks are mapped from (measurement knob; 0: default); blocks of the previous
    size stop serving requests — `trim()` releases the free ones"""
    _ffi.check(_ffi.lib().cnsn_arena_set_chunk_bytes(int(mb * (1 << 20))), "cnsn_arena_set_chunk_bytes")
