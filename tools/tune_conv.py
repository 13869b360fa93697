#!/usr/bin/env python3
"""Times every distinct conv/linear launch of one forward under each workgroup tile (128x128, 128x64,
64x64) and split-K factor, prints the best configuration per shape and the C++ table rows for
madm_amd/csrc/igemm_tuned.inc.  Usage: python tools/tune_conv.py [--dtype bf16] [--reps 20]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--workload", default="extract", choices=["extract", "eval"])
    args = ap.parse_args()
    from madm_amd.ldm_rocm import LdmRocm
    from madm_amd import ops
    from madm_amd._lib import lib, Conv2dArgs
    import bench
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.workload == "eval":
        m = bench.build_eval_model(dtype, torch.device("cuda"))
        call = ([{"target_second_modality": 255.0 * torch.rand((3, args.size, args.size)).cuda()}],)
    else:
        m = LdmRocm("", [], [5, 8, 11], [], input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                    compute_dtype=dtype, weights='synthetic', seed=0)
        call = (bench.make_inputs(args.batch, args.size, torch.device("cuda")), "rgb")
    m(*call)
    torch.cuda.synchronize()
    lib.madm_debug_set_conv_tile(-1)   # heuristic only (ignore the tuned table) while capturing
    ops.CAPTURE = []
    m(*call)
    torch.cuda.synchronize()
    cap = ops.CAPTURE
    ops.CAPTURE = None
    uniq = {}
    for a, keep, desc in cap:
        uniq.setdefault(desc, [a, keep, 0])
        uniq[desc][2] += 1
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    rows = []
    total_base = total_best = 0.0
    for desc, (a0, keep, count) in uniq.items():
        M = a0.B * a0.OH * a0.OW
        K = a0.KH * a0.KW * (a0.C1 + a0.C2)
        nk = K // (64 if args.dtype == "bf16" else 32)
        results = {}
        halo_ok = bool(lib.madm_conv2d_can_fuse_groupnorm(ctypes.byref(a0)))
        tiles = (4, 5) if a0.gn_scale else ((1, 2, 3, 4, 5) if halo_ok else (1, 2, 3))
        nchunks = (a0.C1 + a0.C2) // (64 if args.dtype == "bf16" else 32)
        for tile in tiles:
            for sk in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
                if sk > 1 and (sk > nk // 2):
                    continue
                if tile >= 4 and sk > nchunks:
                    continue
                if sk > 1 and M * a0.N * 4 * sk > ws.numel():
                    continue
                if sk > 1 and M > 8192:
                    continue
                a = Conv2dArgs()
                ctypes.memmove(ctypes.byref(a), ctypes.byref(a0), ctypes.sizeof(a0))
                a.splitk = sk
                a.workspace = ws.data_ptr()
                a.workspace_bytes = ws.numel()
                lib.madm_debug_set_conv_tile(tile)
                for _ in range(3):
                    rc = lib.madm_conv2d_fwd(ctypes.byref(a), stream)
                assert rc == 0, lib.madm_last_error()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    lib.madm_conv2d_fwd(ctypes.byref(a), stream)
                e1.record()
                torch.cuda.synchronize()
                results[(tile, sk)] = e0.elapsed_time(e1) / args.reps * 1e3
        lib.madm_debug_set_conv_tile(-1)
        a = Conv2dArgs()
        ctypes.memmove(ctypes.byref(a), ctypes.byref(a0), ctypes.sizeof(a0))
        base_tile = lib.madm_conv2d_pick_tile(ctypes.byref(a))
        a.splitk = 1
        base_sk = lib.madm_conv2d_suggest_splitk(ctypes.byref(a))
        base = results.get((base_tile, base_sk), float("nan"))
        best = min(results.items(), key=lambda kv: kv[1])
        fl = 2.0 * M * a0.N * K
        total_base += base * count
        total_best += best[1] * count
        rows.append((best[1] * count, desc, count, base_tile, base_sk, base, best[0][0], best[0][1], best[1],
                     fl / best[1] / 1e6, M, a0.N, K, a0.KH))
    lib.madm_debug_set_conv_tile(0)
    rows.sort(reverse=True)
    print(f"heuristic total {total_base / 1e3:.3f} ms  ->  tuned total {total_best / 1e3:.3f} ms")
    print(f"{'shape':40s} {'n':>3s} {'heur':>9s} {'us':>8s} {'best':>9s} {'us':>8s} {'TF/s':>7s}")
    for r in rows:
        print(f"{r[1]:40s} {r[2]:3d} t{r[3]}/sk{r[4]:<3d} {r[5]:8.1f}   t{r[6]}/sk{r[7]:<3d} {r[8]:8.1f} {r[9]:7.1f}")
    print("\n// ---- rows for igemm_tuned.inc: {dtype, M, N, K, KH, tile, splitk}")
    seen = set()
    dt = 1 if args.dtype == "bf16" else 0
    for r in sorted(rows, key=lambda r: (r[10], r[11], r[12], 0 if " gn" in r[1] else 1)):
        key = (r[10], r[11], r[12], r[13])
        if key in seen:
            continue
        seen.add(key)
        print(f"{{{dt}, {r[10]}, {r[11]}, {r[12]}, {r[13]}, {r[6]}, {r[7]}}},")


if __name__ == "__main__":
    main()
