import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madm_amd import ops
B,H,W,C=2,512,512,1024
x=torch.randn((B*H*W,C),device="cuda").to(torch.float16)
dy=torch.randn((B*H*W,C),device="cuda").to(torch.float16)
for kern in ("1","4"):
  os.environ["MADM_DWCONV_KERNEL"]=kern
  for dil in (6,12,18):
      for _ in range(2): ops.dwconv3x3_wgrad(x,dy,B,H,W,dil)
      torch.cuda.synchronize()
      e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(3): dw=ops.dwconv3x3_wgrad(x,dy,B,H,W,dil)
      e1.record(); torch.cuda.synchronize()
      print("wgrad kernel",kern,"dil",dil,e0.elapsed_time(e1)/3*1e3,"us", float(dw.abs().sum()))
