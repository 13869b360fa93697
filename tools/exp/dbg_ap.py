import math, sys, torch
sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
from madm_amd import ops, packing
from madm_amd._lib import lib
def gen(shape, seed): return torch.randn(shape, generator=torch.Generator().manual_seed(seed))
dtype = torch.float16
for (M, C, N) in ((64, 320, 64), (64, 320, 128), (1000, 320, 64), (1000, 320, 2560)):
    kt = ops.k_tile(dtype)
    x = gen((M, C), 1) * (0.5 + 2.0 * torch.rand((M, 1), generator=torch.Generator().manual_seed(7))) + 3.0 * gen((M, 1), 8)
    x = x.to(dtype).float()
    gamma, beta = torch.ones(C), torch.zeros(C)
    w = torch.zeros((N, C)); 
    for n in range(N): w[n, n % C] = 1.0          # out[m][n] = LN(x)[m][n % C]
    b = torch.zeros(N)
    ref = F.linear(F.layer_norm(x, (C,), gamma, beta, 1e-5), w, b)
    lib.madm_debug_set_conv_tile(13)
    wp, bp, cs = packing.fold_layernorm(w, b, gamma, beta, dtype, kt)
    for rep in range(3):
        out = ops.linear(x.to(dtype).cuda(), wp.cuda(), bias=bp.cuda(), ln=(cs.cuda(), 1e-5))
        torch.cuda.synchronize()
        d = (out.float().cpu() - ref).abs()
        bad = (d > 0.02).nonzero()
        rows = sorted(set(bad[:, 0].tolist()))
        print(M, C, N, "rep", rep, "bad", bad.shape[0], "rows", rows[:10], "cols", sorted(set(bad[:, 1].tolist()))[:16])
        if rows:
            r = rows[0]; cs_ = sorted(set(bad[bad[:, 0] == r][:, 1].tolist()))
            print("   row", r, "x mean/std", x[r].mean().item(), x[r].std().item(), "out", out[r, cs_[:6]].float().cpu().tolist(), "ref", ref[r, cs_[:6]].tolist(), "x", [x[r, c % C].item() for c in cs_[:6]])
