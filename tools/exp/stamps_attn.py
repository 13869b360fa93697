#!/usr/bin/env python3
"""Per-KV-tile stamps of the attention kernel (ATTN_STAMPS build), block 0 / wave 0:
[0] tile start (next tile's loads issued after it), [1] S = K Q^T issued, [2] soft-max done, [3] O += V P issued,
[4] next tile written to LDS, [5] barrier passed."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madm_amd import ops
L, D = int(sys.argv[1]), int(sys.argv[2])
B, H = 2, 8
qkv = torch.randn((B * L, 3 * H * D), device="cuda").to(torch.bfloat16)
q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
for _ in range(3):
    ops.attention(q, k, v, B, H, L, L, D, D ** -0.5)
torch.cuda.synchronize()
fn = ctypes.CDLL(os.environ["MADM_HIP_LIB"]).madm_debug_read_attn_stamps
buf = (ctypes.c_ulonglong * 1024)()
assert fn(buf, 1024) == 0
n = min(128, (L + 127) // 128)
print("tile:  loads+QK  softmax     PV   store  barrier | total")
tot = [0] * 5
for t in range(n - 1):
    s = [buf[t * 8 + i] for i in range(6)]
    d = [s[i + 1] - s[i] for i in range(5)]
    tot = [a + b for a, b in zip(tot, d)]
    if t < 6:
        print(f"{t:4d}: {d[0]:8d} {d[1]:8d} {d[2]:6d} {d[3]:7d} {d[4]:8d} | {s[5] - s[0]}")
print("avg :", [x // (n - 1) for x in tot], "per tile", sum(tot) // (n - 1))
