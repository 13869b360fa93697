import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madm_amd import ops
B,H,W,C=2,512,512,1024
x=torch.randn((B*H*W,C),device="cuda").to(torch.float16)
w=torch.randn((9,C),device="cuda"); one=torch.ones(C,device="cuda"); zero=torch.zeros(C,device="cuda")
for dil in (6,12,18):
    for _ in range(2): ops.dwconv3x3(x,w,one,zero,B,H,W,dil,0)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.dwconv3x3(x,w,one,zero,B,H,W,dil,0)
    e1.record(); torch.cuda.synchronize()
    print("dil",dil,e0.elapsed_time(e1)/5*1e3,"us")
