#!/bin/bash
# Tuning builds: the two SelfNorm-cluster translation units recompiled with extra -D flags and linked with the other
# objects of the regular build into scratch/lib_<name>.so (git-ignored; use with CNSN_LIB_PATH=...).
#   tools/build_variant.sh ppw8 -DSNX_NV1_PPW8=8
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/crossnorm-selfnorm_amd/csrc
out=$root/scratch; mkdir -p $out
for u in cnsn_resident_sn cnsn_resident_sn_bwd; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $csrc/$u.hip -o $out/${u}_$name.o &
done
wait
objs=$(ls $csrc/*.o | grep -v "cnsn_resident_sn\(_bwd\)\?\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $out/cnsn_resident_sn_$name.o $out/cnsn_resident_sn_bwd_$name.o -o $out/lib_$name.so
rm -f $out/*_$name.o
echo $out/lib_$name.so
