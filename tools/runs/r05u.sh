cat > /tmp/u.py <<'PY'
import torch, numpy as np, os
from cnsn_amd.callers import ResNet50CNSN
CL=torch.channels_last; DEV="cuda:0"
for seed in (4,5):
    torch.manual_seed(seed)
    a = ResNet50CNSN(num_classes=10, cnsn_type="sn", pos="post").to(DEV).train()
    b = ResNet50CNSN(num_classes=10, cnsn_type="sn", pos="post").to(DEV); b.load_state_dict(a.state_dict()); b=b.to(memory_format=CL).train()
    x = torch.randn(6,3,96,96,device=DEV); y=torch.randint(0,10,(6,),device=DEV)
    outs=[]
    for m,xx in ((a,x),(b,x.contiguous(memory_format=CL))):
        l=m(xx); torch.nn.functional.cross_entropy(l,y).backward(); outs.append(l.detach())
    torch.cuda.synchronize()
    def rel(u,v): return float((u-v).abs().max())/max(float(u.abs().max()),1e-12)
    sa,sb=[m.layer1[0].cnsn.selfnorm for m in (a,b)]
    print(os.environ.get("CNSN_NHWC"),"logits",rel(outs[0],outs[1]),"conv1 grad",rel(a.conv1.weight.grad,b.conv1.weight.grad),"gate grad",rel(sa.g_fc.weight.grad,sb.g_fc.weight.grad))
PY
PYTHONPATH=$PWD CNSN_NHWC=1 python /tmp/u.py 2>&1 | tail -4
PYTHONPATH=$PWD CNSN_NHWC=0 python /tmp/u.py 2>&1 | tail -4
