#!/usr/bin/env python3
"""Self-attention shapes of the UNet in a hipGraph (bf16, fused-QKV layout as the model uses it)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madm_amd import ops
for (L, D) in ((4096, 40), (1024, 80), (256, 160)):
    B, H = 2, 8
    qkv = torch.randn((B * L, 3 * H * D), device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    f = lambda: ops.attention(q, k, v, B, H, L, L, D, D ** -0.5)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"attn B{B} H{H} L{L} D{D}: {us:.1f} us  {4.0 * B * H * L * L * D / us / 1e6:.1f} TF/s")
