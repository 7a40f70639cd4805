"""Where do the op's outputs lie, and how fast are they written there?  (measurement aid; results: profiles/r05_arena.md)

One process, one x of the north-star shape.  Write targets, INTERLEAVED in time so that every kind draws its physical memory
from the same parts of the device as the others:
  plain        torch.empty_like (one hipMalloc'ed block of the tensor's size from the caching allocator)
  arena <MiB>  cnsn_amd.arena blocks mapped from physical chunks of that size
Probe = one inference-mode SelfNorm launch x -> out (cnsn_amd.placement._Probe: the cluster kernels' own plane-strided access
order), ms per launch and TB/s on 2*E*b.  Every block stays alive until the end, so no two targets share memory.
usage: python tools/arena_probe.py [rounds=4] [chunk sizes in MiB, comma separated]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd  # noqa: E402
from cnsn_amd import arena, placement  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    chunks = [float(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2,8,16,32,56,64,128,196,256,784").split(",")]
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if os.environ.get("PROBE_DTYPE") == "bf16" else torch.float32
    shape = tuple(int(v) for v in os.environ.get("PROBE_SHAPE", "256,256,56,56").split(","))
    x = torch.randn(shape, device=dev).to(dt)
    nbytes = x.numel() * x.element_size()
    probe = placement._Probe(x)
    keep, rows = [], {}
    for r in range(rounds):
        t = torch.empty_like(x)
        keep.append(t)
        rows.setdefault("plain", []).append(probe.ms(t))
        for mb in chunks:
            arena.set_chunk_mb(mb)
            t = arena.empty_like(x)
            assert cnsn_amd.lib().cnsn_arena_owns(t.data_ptr()) == 1, "arena fell back to torch's allocator"
            keep.append(t)
            rows.setdefault(f"arena {mb:g} MiB", []).append(probe.ms(t))
    # the arena's own fill probe (cnsn_arena_prospect) against the library's launch, block by block
    arena.set_chunk_mb(0)
    import ctypes as C
    n = 12
    rates = (C.c_float * n)()
    kept = cnsn_amd.lib().cnsn_arena_prospect(0, nbytes, n, n, C.c_void_p(torch.cuda.current_stream().cuda_stream), rates)
    blocks = [arena.empty_like(x) for _ in range(kept)]
    pairs = sorted(((arena.block_gbps(t), 2 * nbytes / probe.ms(t) / 1e9) for t in blocks), reverse=True)
    print("\nfill probe GB/s vs the library's launch GB/s on the same block (prospect, 12 blocks):")
    print(" ".join(f"{a:.0f}/{b * 1000:.0f}" for a, b in pairs))
    keep += blocks
    torch.cuda.synchronize()
    print(f"| target | ms per launch (x -> out, {shape} {dt}) | TB/s on 2*E*b |")
    print("|---|---|---|")
    for k, v in rows.items():
        print(f"| {k} | {' '.join(f'{m:.4f}' for m in v)} | {' '.join(f'{2 * nbytes / m / 1e9:.2f}' for m in v)} |")
    print(json.dumps({"arena": arena.stats(), "rows": rows}))


if __name__ == "__main__":
    main()
