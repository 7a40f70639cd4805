"""The build's own guards (no GPU needed: hipcc cross-compiles gfx950 and the checks read machine code).

* `__graft_entry__.check_store_hazard`: no buffer store of more than 8 bytes may take its offset from an SGPR — on gfx950
  the next VALU instruction can overwrite the store's data registers before the store has read them, and hipcc pads that
  hazard only for the immediate-soffset form (tools/store_hazard_probe.hip).  The check must refuse a
  translation unit that contains such a store and accept the library's own objects."""
import glob
import os
import shutil
import subprocess

import pytest

import __graft_entry__ as entry

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

BAD = r"""
#include <hip/hip_runtime.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* dst, int bytes, int soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, bytes, 0x00020000);
    const int so = __builtin_amdgcn_readfirstlane(soff);
    v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, so, 0);
    v.x += 1;
    __builtin_amdgcn_raw_buffer_store_b128(v, w, threadIdx.x * 16, SOFFSET, 0);
}
"""


def _compile(tmp_path, name, soffset):
    src = tmp_path / f"{name}.hip"
    src.write_text(BAD.replace("SOFFSET", soffset))
    obj = tmp_path / f"{name}.o"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", str(obj)], stderr=subprocess.DEVNULL)
    return str(obj)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_store_hazard_check_refuses_a_register_soffset(tmp_path):
    with pytest.raises(AssertionError, match="register soffset"):
        entry.check_store_hazard(_compile(tmp_path, "bad", "so"))
    assert entry.check_store_hazard(_compile(tmp_path, "good", "0")) == 1     # the same store, offset 0: counted, accepted


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_the_librarys_own_objects_pass():
    objs = sorted(glob.glob(os.path.join(entry.CSRC, "*.o")))
    if not objs:
        pytest.skip("the library has not been built in this tree")
    # (one of the cluster-kernel units is enough here: the build itself checks all of them)
    unit = [o for o in objs if o.endswith("cnsn_resident_sn.o")] or objs[:1]
    assert entry.check_store_hazard(unit[0]) > 0
