#!/usr/bin/env python3
"""tools/isa_loops.py <object with a gfx950 bundle> <kernel-name regex>: where a kernel's instructions are — per loop.

The static counts of tools/isa_stats.sh cannot say what an ITEM of a persistent kernel executes (round-5 review, item 4: "899
lane moves — on the item loop or not?").  This script disassembles the kernel, finds its loops from the backward branches
(a branch to an earlier address closes a loop; loops nest by containment), and prints for every loop its span, its depth and the
instructions by class that lie in it but in none of its inner loops.  The item loop of the cluster kernels is the OUTERMOST loop
with the largest body; its own count (plus the bodies of the short loops inside it, once each, which is what an iteration without
waiting executes) is the dynamic count per item.  Measurement aid; no GPU needed."""
import re
import subprocess
import sys
import tempfile

obj, pat = sys.argv[1], re.compile(sys.argv[2])
tmp = tempfile.mkdtemp()
subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, tmp + "/fat.bin"])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--input=" + tmp + "/fat.bin",
                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + tmp + "/dev.co", "--unbundle"])
dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", tmp + "/dev.co"], capture_output=True, text=True, check=True).stdout

CLASSES = (("lane", lambda o: o.startswith(("v_readlane", "v_writelane", "v_readfirstlane"))),
           ("dpp", lambda o, l="": "dpp" in l or "row_" in l),
           ("valu", lambda o: o.startswith("v_")),
           ("salu", lambda o: o.startswith("s_") and not o.startswith(("s_load", "s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch", "s_buffer"))),
           ("smem", lambda o: o.startswith(("s_load", "s_buffer_load"))),
           ("vmem", lambda o: o.startswith(("buffer_", "global_", "flat_"))),
           ("lds", lambda o: o.startswith("ds_")),
           ("wait", lambda o: o.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep"))))

kernels = {}
cur = None
for line in dis.splitlines():
    m = re.match(r"^([0-9a-f]+) <(.+)>:", line)
    if m:
        cur = m.group(2)
        kernels[cur] = []
        continue
    if cur is None:
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
    if m:
        kernels[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))

names = [n for n in kernels if kernels[n]]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for name, d in zip(names, dem):
    short = re.sub(r"\(.*", "", d).replace("void cnsn::", "")
    if not pat.search(short):
        continue
    ins = kernels[name]
    addr_idx = {a: i for i, (a, _, _) in enumerate(ins)}
    loops = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            try:                                   # the operand is a signed 16-bit count of dwords behind the branch
                imm = int(args.split()[0])
            except (ValueError, IndexError):
                continue
            if imm >= 32768:
                imm -= 65536
            tgt = a + 4 + 4 * imm
            if tgt is not None and tgt in addr_idx and addr_idx[tgt] <= i:
                loops.append((addr_idx[tgt], i))
    # merge loops with the same head (several back edges): keep the widest
    heads = {}
    for h, t in loops:
        heads[h] = max(heads.get(h, t), t)
    loops = sorted(heads.items(), key=lambda x: (x[0], -x[1]))

    def depth(l):
        return sum(1 for o in loops if o != l and o[0] <= l[0] and l[1] <= o[1])

    def own(l):   # instruction indices in l but in none of the loops inside it
        inner = [o for o in loops if o != l and l[0] <= o[0] and o[1] <= l[1]]
        return [i for i in range(l[0], l[1] + 1) if not any(o[0] <= i <= o[1] for o in inner)]

    def count(idx):
        c = {k: 0 for k, _ in CLASSES}
        for i in idx:
            _, op, args = ins[i]
            if "dpp" in args or "row_" in args or "quad_perm" in args:
                c["dpp"] += 1
            for k, f in CLASSES:
                if k == "dpp":
                    continue
                if f(op):
                    c[k] += 1
                    break
        return c
    total = count(range(len(ins)))
    print(f"== {short}: {len(ins)} instructions  " + " ".join(f"{k} {v}" for k, v in total.items()))
    inloop = set()
    for l in loops:
        inloop.update(range(l[0], l[1] + 1))
    c0 = count([i for i in range(len(ins)) if i not in inloop])
    print(f"   outside every loop: {len(ins) - len(inloop):5d}  " + " ".join(f"{k} {v}" for k, v in c0.items()))
    for l in loops:
        idx = own(l)
        c = count(idx)
        whole = l[1] - l[0] + 1
        print(f"   {'  ' * depth(l)}loop @{l[0]:5d}..{l[1]:5d} (body {whole:5d}, own {len(idx):5d})  " + " ".join(f"{k} {v}" for k, v in c.items()))
