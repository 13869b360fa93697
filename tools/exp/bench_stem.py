import torch, sys, os
sys.path.insert(0, os.getcwd())
from madm_amd import ops
img = torch.rand((2, 3, 512, 512), device="cuda")
wT = torch.randn((27, 128), device="cuda"); b = torch.randn(128, device="cuda")
st = torch.zeros((2, 128, 2), dtype=torch.float64, device="cuda")
for _ in range(3): ops.stem_conv3x3(img, wT, b, torch.float16, 0.5, 0.5, stats=st)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ops.stem_conv3x3(img, wT, b, torch.float16, 0.5, 0.5, stats=st)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20): ops.stem_conv3x3(img, wT, b, torch.float16, 0.5, 0.5, stats=st)
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
print(os.environ.get("MADM_HIP_LIB", "default"), "stem conv 2x512x512:", round(e0.elapsed_time(e1) * 1e3 / 20, 1), "us per launch")
