"""write rate of arena candidates in the order they were created: small blocks (one chunk each) against 784 MiB blocks"""
import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
import cnsn_amd
from cnsn_amd import _ffi
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)
lib = _ffi.lib()
MiB = 1 << 20
def sweep(nbytes, n, tag, per_line=56):
    rates = (C.c_float * n)()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    k = lib.cnsn_arena_prospect(0, nbytes, 0, n, st, rates)
    v = [x / 1000 for x in rates if x > 0]
    print(tag, "n", len(v), "min %.2f median %.2f max %.2f" % (min(v), sorted(v)[len(v) // 2], max(v)))
    for i in range(0, len(v), per_line):
        print("   ", " ".join(f"{x:.1f}" for x in v[i:i + per_line]))
    return v
sweep(784 * MiB, 28, "784 MiB blocks (14 chunks of 56 MiB):")
sweep(56 * MiB, 392, "56 MiB blocks (one chunk): the same 22 GB")
sweep(784 * MiB, 28, "784 MiB blocks again:")
sweep(196 * MiB, 112, "196 MiB blocks (4 chunks... of 56 -> 224):")
