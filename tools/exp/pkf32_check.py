#!/usr/bin/env python3
"""Packed-FP32 observation of DESIGN.md 9.3, instrumented.  Needs the AP_PKCHECK build of
igemm_apanel.hip (build/libmadm_hip_pkcheck.so: the A-stationary kernel WITH packed-FP32 ops; every normalised element is
re-derived by scalar v_sub_f32 / v_mul_f32 from the same registers inside the kernel, mismatches are recorded):

    MADM_HIP_LIB=build/libmadm_hip_pkcheck.so python tools/exp/pkf32_check.py [--reps 40]
    MADM_APANEL_LDS_PAD=90000 MADM_HIP_LIB=... python tools/exp/pkf32_check.py      (control: one workgroup per CU)

Prints, per shape: wrong output elements against a torch reference, the in-kernel mismatch count, and the decoded records
(lane, piece round u, EXEC mask, operand and result bits, hardware id)."""
import argparse
import ctypes
import os
import struct
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madm_amd import ops, packing  # noqa: E402
from madm_amd._lib import lib, LIB_PATH  # noqa: E402


def f32(u):
    return struct.unpack("f", struct.pack("I", u))[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--dtype", default="f16")
    args = ap.parse_args()
    dtype = {"f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    raw = ctypes.CDLL(LIB_PATH)
    has = hasattr(raw, "madm_debug_read_ap_pkcheck")
    print("lib", LIB_PATH, "pkcheck export:", has, "LDS pad", os.environ.get("MADM_APANEL_LDS_PAD", "0"))
    buf = (ctypes.c_uint * 1024)()
    if has:
        raw.madm_debug_read_ap_pkcheck(buf, 1024, 1)
    g = torch.Generator().manual_seed(1)
    for (M, C, N) in ((8192, 320, 960), (8192, 320, 2560), (2048, 640, 1920), (512, 1280, 3840), (1000, 320, 2560)):
        kt = ops.k_tile(dtype)
        x = torch.randn((M, C), generator=g) * (0.5 + 2.0 * torch.rand((M, 1), generator=g)) + 3.0 * torch.randn((M, 1), generator=g)
        x = x.to(dtype)
        gamma, beta = torch.ones(C), torch.zeros(C)
        w = torch.randn((N, C), generator=g) / C ** 0.5
        b = torch.randn(N, generator=g)
        wp, bp, cs = packing.fold_layernorm(w, b, gamma, beta, dtype, kt)
        xd, wd, bd, cd = x.cuda(), wp.cuda(), bp.cuda(), cs.cuda()
        ref = F.linear(F.layer_norm(xd.float(), (C,), None, None, 1e-5).to(dtype).float(), wd.float(), bd)
        lib.madm_debug_set_conv_tile(13)
        bad_tot, mism_tot = 0, 0
        first = None
        hist = {}
        try:
            for rep in range(args.reps):
                out = ops.linear(xd, wd, bias=bd, ln=(cd, 1e-5))
                torch.cuda.synchronize()
                d = (out.float() - ref).abs()
                bad = int((d > 0.05 + 0.02 * ref.abs()).sum())
                bad_tot += bad
                if has:
                    raw.madm_debug_read_ap_pkcheck(buf, 1024, 1)
                    mism_tot += buf[0]
                    recs = [list(buf[16 + 16 * k: 32 + 16 * k]) for k in range(min(buf[0], 60))]
                    for r in recs:      # (element, 16-lane group, piece round, wave slot on its SIMD, active lanes of the mismatch)
                        key = (r[4], (r[1] & 63) >> 4, r[3], r[12] & 15, f"{r[11]:08x}{r[10]:08x}")
                        hist[key] = hist.get(key, 0) + 1
                    if buf[0] and first is None:
                        first = recs[:4]
        finally:
            lib.madm_debug_set_conv_tile(0)
        print(f"M{M} K{C} N{N}: {args.reps} launches, wrong output elements {bad_tot}, in-kernel packed != scalar {mism_tot}")
        for key, n in sorted(hist.items()):
            print(f"   records: element {key[0]} lane group {key[1]} (lanes {16 * key[1]}..{16 * key[1] + 15}) piece round u={key[2]} wave slot {key[3]} mismatching lanes {key[4]}: {n}")
        for r in first or []:
            hw = r[12]
            print(f"   wg {r[0]:5d} tid {r[1]:3d} (lane {r[1] & 63:2d}, l16 {r[1] & 15:2d}) row {r[2]:3d} u {r[3]} e {r[4]}  x {f32(r[5]):9.5f} mean {f32(r[6]):9.5f} "
                  f"rstd {f32(r[7]):8.5f}  packed {f32(r[8]):9.5f} scalar {f32(r[9]):9.5f}  x*rstd {f32(r[5]) * f32(r[7]):9.5f}  "
                  f"exec {r[11]:08x}{r[10]:08x}  hw_id {hw:08x} (wave {hw & 15}, simd {(hw >> 4) & 3}, cu {(hw >> 8) & 15}, sh {(hw >> 12) & 1}, se {(hw >> 13) & 7})  pieces {r[13]}")


if __name__ == "__main__":
    main()
