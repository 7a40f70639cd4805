"""Where the op's OUTPUT tensors lie in device memory decides how fast they are written.

Measured on MI355X (profiles/r04_placement_sensitivity.md, profiles/r04_memory_map.md): device memory consists of regions of two
kinds, tens of GB each, stable for the life of a box.  Plane-strided writes — what every single-touch launch of this library
does to y (forward) and dx (backward) — run at 6.2 TB/s (copy rate) into one kind and at 5.1-5.4 TB/s into the other, whatever
is being read; about one 800 MB allocation in five or six lies in a fast region.  Neither the virtual address, nor the store's
cache policy, nor the order of the accesses changes that, and the driver offers no way to ask for a region: all that user
space can do is LOOK — write into a block and time it.

`prefer_fast_write_blocks(like)` does that for torch's caching allocator: it takes `candidates` blocks of `like`'s size from the
allocator, times a plane-strided write into each with one of the library's own launches (SelfNorm in inference mode: x in, y
out, no exchange, no state), returns the slow ones to the DRIVER (`torch.cuda.empty_cache()`) and the `keep` fastest to the
allocator's free list — which hands out its free blocks of a size before it asks the driver for new ones, so the next
tensors of that size (the op's y and dx, allocated per call and freed per step) are written where writing is fast.  A training
job calls it once per large activation size after building the model; nothing in the library depends on it.

The allocator is told not to split blocks of this size for smaller requests (`max_split_size_mb`), otherwise the first small
allocation would carve up a kept block.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _ffi
from . import functional as _F

__all__ = ["probe_write_ms", "prefer_fast_write_blocks"]


class _Probe:
    """one inference-mode SelfNorm launch x -> out through the C ABI (the output pointer is ours to choose there)"""

    def __init__(self, x: torch.Tensor):
        _F._require_device(x, "placement probe")
        self.x = x
        c = int(x.shape[1])
        dev = x.device
        self.cfg = _F.FusedConfig(sn_active=True, sn_training=False)
        self.prob = _F._problem(x, self.cfg)
        _F._context(self.prob, dev)
        self.gate = _F._GateBuffers(torch.full((c, 1, 2), 0.1, device=dev), torch.ones(c, device=dev), torch.zeros(c, device=dev),
                                    torch.zeros(c, device=dev), torch.ones(c, device=dev))
        ws_bytes = _F._sizes(self.prob)[1]
        self.ws = torch.empty(ws_bytes // 4 + 1, dtype=torch.float32, device=dev)
        self.ws_bytes = ws_bytes
        self.lib = _ffi.lib()

    def launch(self, out: torch.Tensor) -> int:
        return self.lib.cnsn_forward_fused(C.byref(self.prob), None, _F._ptr(self.x), None, None, C.byref(self.gate.c), None,
                                           _F._ptr(out), None, _F._ptr(self.ws), self.ws_bytes, _F._stream(self.x))

    def ms(self, out: torch.Tensor, launches: int = 6) -> float:
        for _ in range(2):
            _ffi.check(self.launch(out), "placement probe")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            self.launch(out)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / launches


def probe_write_ms(x: torch.Tensor, out: torch.Tensor, launches: int = 6) -> float:
    """ms per plane-strided x -> out launch (what `prefer_fast_write_blocks` ranks blocks by)"""
    assert out.shape == x.shape and out.dtype == x.dtype and out.is_contiguous()
    return _Probe(x).ms(out, launches)


def prefer_fast_write_blocks(like: torch.Tensor, keep: int = 4, candidates: int = 64, min_gain: float = 0.05,
                             max_bytes: Optional[int] = None) -> dict:
    """Leave the caching allocator with `keep` free blocks of `like`'s size that lie where plane-strided writes are fast.

    like        a dense (N, C, H, W) device tensor of the size and dtype in question (read by the probe, not modified)
    candidates  blocks looked at (each `like.nbytes`; all but `keep` go back to the driver); max_bytes caps their total
    min_gain    blocks are only kept if the fastest is at least this much faster than the median: otherwise this part of the
                device memory is uniform (or uniformly slow) and the allocator is left as it was
    Returns a report: probe times, what was kept.  Costs candidates x ~10 launches; call it once, outside timed regions."""
    x = like.detach()
    nbytes = x.numel() * x.element_size()
    if max_bytes is not None:
        candidates = max(keep + 1, min(candidates, int(max_bytes // max(1, nbytes))))
    free_b, _total = torch.cuda.mem_get_info(x.device)
    candidates = max(0, min(candidates, int(free_b * 0.8 // max(1, nbytes))))
    report = {"bytes_per_block": nbytes, "candidates": candidates, "kept": 0, "probe_ms": None}
    if candidates <= keep:
        return report
    try:   # blocks of this size are not to be split for smaller requests (and small requests are not served from them)
        setting = f"max_split_size_mb:{max(32, min(512, (nbytes >> 20) // 2))}"
        setter = getattr(torch._C, "_accelerator_setAllocatorSettings", None)
        if setter is not None:
            setter(setting)
        else:
            torch.cuda.memory._set_allocator_settings(setting)
        report["allocator"] = "max_split_size_mb set"
    except Exception as e:  # an allocator that does not know the option: the kept blocks may get carved up
        report["allocator"] = f"max_split_size_mb not set ({type(e).__name__})"
    probe = _Probe(x)
    blocks = []
    try:
        for _ in range(candidates):
            blocks.append(torch.empty_like(x))
    except torch.OutOfMemoryError:
        pass
    times = [probe.ms(b) for b in blocks]
    torch.cuda.synchronize(x.device)
    order = sorted(range(len(blocks)), key=lambda i: times[i])
    srt = [times[i] for i in order]
    med = srt[len(srt) // 2]
    report["probe_ms"] = {"min": round(srt[0], 4), "median": round(med, 4), "max": round(srt[-1], 4),
                          "kept_max": None, "fast_blocks": sum(1 for t in srt if t <= med * (1.0 - min_gain))}
    good = [i for i in order[:keep] if times[i] <= med * (1.0 - min_gain)]
    kept = [blocks[i] for i in good]
    report["kept"] = len(kept)
    report["kept_ptrs"] = [hex(t.data_ptr()) for t in kept]
    if kept:
        report["probe_ms"]["kept_max"] = round(max(times[i] for i in good), 4)
    del blocks, probe
    torch.cuda.synchronize(x.device)
    torch.cuda.empty_cache()          # every free cached block back to the driver — the kept ones are still referenced
    del kept                          # ... and now become the allocator's only free blocks of this size
    return report
