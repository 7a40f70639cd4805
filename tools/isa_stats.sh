#!/bin/bash
# usage: tools/isa_stats.sh <object with a gfx950 bundle>  — per kernel: instructions, lane moves, scratch ops, s_nop
set -e
obj=$1; tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $obj $tmp/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-objdump -d $tmp/dev.co > $tmp/dev.s
python3 - $tmp/dev.s <<'PY'
import re,sys,subprocess
cur=None; st={}
for line in open(sys.argv[1]):
    m=re.match(r"^[0-9a-f]+ <(.+)>:",line)
    if m: cur=m.group(1); st[cur]=dict(n=0,lane=0,scratch=0,nop=0,valu=0,vmem=0); continue
    if cur is None: continue
    t=line.split()
    if not t or t[0].endswith(':'): continue
    op=t[0]
    if not re.match(r"^[a-z]",op): continue
    s=st[cur]; s['n']+=1
    if op.startswith(('v_readlane','v_writelane')): s['lane']+=1
    elif op.startswith('scratch_'): s['scratch']+=1
    elif op=='s_nop': s['nop']+=1
    if op.startswith('v_'): s['valu']+=1
    if op.startswith(('buffer_','global_')): s['vmem']+=1
names=list(st)
dem=subprocess.run(['c++filt'],input="\n".join(names),capture_output=True,text=True).stdout.splitlines()
for n,d in zip(names,dem):
    d=re.sub(r"\(.*","",d).replace("void cnsn::","")
    s=st[n]
    print(f"{d[:86]:86s} instr {s['n']:5d} valu {s['valu']:5d} lane-moves {s['lane']:4d} scratch {s['scratch']:3d} s_nop {s['nop']:4d} vmem {s['vmem']:3d}")
PY
rm -rf $tmp
