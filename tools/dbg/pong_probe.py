"""Is tests/test_gpu_pipe.py::test_granule_regions_* really on the granule regions?  Runs its call sequence once with CNSN_PONG
from the environment; under `rocprofv3 --kernel-trace --stats` the number of fill launches tells (tools/dbg: a measurement aid)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cnsn_amd  # noqa: E402
from tests import test_gpu_pipe as t  # noqa: E402

cnsn_amd.set_strategy("resident")
os.environ["CNSN_SNX"] = "0"
cnsn_amd.reload_env()
t._pong_sequence(sys.argv[1] if len(sys.argv) > 1 else "2")
torch.cuda.synchronize()
print("done")
