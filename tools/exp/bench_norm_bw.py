"""Streaming norm kernels of the training step against their algorithmic bytes (HBM-bound rows of DESIGN section 3):
GroupNorm / BatchNorm statistics, apply, backward on the VAE-decoder and head tensor sizes."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madm_amd import ops

def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

dt = torch.float16
for (B, HW, C, G, kind) in ((2, 512 * 512, 128, 32, "gn"), (2, 256 * 256, 256, 32, "gn"), (2, 64 * 64, 320, 32, "gn"),
                            (1, 2 * 512 * 512, 256, 256, "bn"), (1, 2 * 512 * 512, 1024, 1024, "bn")):
    M = B * HW
    x = torch.randn((M, C), device="cuda").to(dt)
    dy = torch.randn((M, C), device="cuda").to(dt)
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    st = torch.zeros((B, C, 2), dtype=torch.float64, device="cuda")
    nbytes = M * C * 2
    def stats():
        st.zero_(); ops.groupnorm_stats(x, B, HW, st)
    t = timed(stats); print(f"{kind} B{B} HW{HW} C{C}: stats          {t:8.1f} us  {nbytes / t / 1e6:6.2f} TB/s (1 read)")
    stats()
    if kind == "gn":
        f = lambda: ops.groupnorm([x], B, HW, G, gamma, beta, 1e-5, silu=True, stats=[st])
        try:
            t = timed(f); print(f"{kind} B{B} HW{HW} C{C}: apply          {t:8.1f} us  {2 * nbytes / t / 1e6:6.2f} TB/s (read + write)")
        except Exception as e:
            print("apply skipped:", str(e)[:80])
    g = lambda: ops.groupnorm_backward([x], dy, B, HW, G, gamma, beta, 1e-5, [st], act="silu" if kind == "gn" else "relu")
    t = timed(g); print(f"{kind} B{B} HW{HW} C{C}: backward (2 k.) {t:8.1f} us  {5 * nbytes / t / 1e6:6.2f} TB/s (x, dy read twice, dx written)")
    del x, dy
