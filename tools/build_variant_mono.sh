#!/bin/bash
# as tools/build_variant.sh, for the channel-in-registers (mono) translation units
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/crossnorm-selfnorm_amd/csrc
out=$root/scratch; mkdir -p $out
for u in cnsn_mono cnsn_mono_tail; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $csrc/$u.hip -o $out/${u}_$name.o &
done
wait
objs=$(ls $csrc/*.o | grep -v "cnsn_mono\(_tail\)\?\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $out/cnsn_mono_$name.o $out/cnsn_mono_tail_$name.o -o $out/lib_$name.so
rm -f $out/*_$name.o
echo $out/lib_$name.so
