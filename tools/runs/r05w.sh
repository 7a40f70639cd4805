python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python - <<'PY'
import time
t=time.time(); import torch; print("import torch s", round(time.time()-t,2))
t=time.time(); import cnsn_amd; cnsn_amd.lib(); print("dlopen s", round(time.time()-t,3))
import cnsn_amd.functional as F
from cnsn_amd import SelfNorm
m=SelfNorm(64).cuda().train()
x=torch.randn(8,64,32,32,device="cuda:0",requires_grad=True)
torch.cuda.synchronize(); t=time.time(); y=m(x); y.sum().backward(); torch.cuda.synchronize(); print("first call s", round(time.time()-t,3))
t=time.time(); y=m(x); y.sum().backward(); torch.cuda.synchronize(); print("second call s", round(time.time()-t,4))
PY
python -m pytest tests/test_gpu_parity.py tests/test_gpu_arena.py -q -m gpu -x 2>&1 | tail -2
