#!/usr/bin/env python3
"""In-situ tile / split-K tuner: runs WHOLE eager forwards with one (tile, split-K) candidate forced on every
eligible conv / linear launch and HIP events around each launch (ops.PROFILE), so every layer is timed with its
real inputs and COLD weights (the 1.7 GB of parameters stream from HBM once per forward; a per-layer loop that
re-launches one layer measures L2-resident weights instead and mis-ranks the latency-bound small-M GEMMs).
Prints the best candidate per shape and the rows of madm_amd/csrc/igemm_tuned.inc.
Usage: python tools/tune_insitu.py [--workload extract|eval] [--reps 3]"""
import argparse
import collections
import os
import re
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TILE_OF = {"igemm_128x128": 1, "igemm_128x64": 2, "igemm_64x64": 3, "conv3x3_halo_x128": 4, "conv3x3_halo_x64": 5,
           "igemm_64x64d": 6, "igemm_glds_64x64": 7, "igemm_glds_128x64": 8,
           "conv3x3_halo_dma_x128": 9, "conv3x3_halo_dma_x64": 10, "igemm_glds_64x64s": 11, "conv3x3_h16_x128": 12,
           "igemm_apanel": 13}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--workload", default="extract", choices=["extract", "eval"])
    ap.add_argument("--tiles", type=int, nargs="*", default=[1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13])
    ap.add_argument("--splitk", type=int, nargs="*", default=[1, 2, 3, 4, 6, 8, 12, 16, 24])
    args = ap.parse_args()
    from madm_amd.ldm_rocm import LdmRocm
    from madm_amd import ops
    from madm_amd._lib import lib
    import bench
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    if args.workload == "eval":
        m = bench.build_eval_model(dtype, torch.device("cuda"))
        call = ([{"target_second_modality": 255.0 * torch.rand((3, args.size, args.size)).cuda()}],)
    else:
        m = LdmRocm("", [], [5, 8, 11], [], input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                    compute_dtype=dtype, weights='synthetic', seed=0)
        call = (bench.make_inputs(args.batch, args.size, torch.device("cuda")), "rgb")
    m(*call)
    torch.cuda.synchronize()

    def profiled():
        ops.PROFILE = []
        m(*call)
        torch.cuda.synchronize()
        rec, ops.PROFILE = ops.PROFILE, None
        return [(name, desc, e0.elapsed_time(e1) * 1e3) for name, _, e0, e1, desc, _, _ in rec if not name.startswith(("attn", "stem_"))]

    # results[launch index][(tile, sk)] = [us...]; the launch sequence is identical in every forward
    base = profiled()
    n = len(base)
    results = [collections.defaultdict(list) for _ in range(n)]

    def run(tile, sk, tag):
        lib.madm_debug_set_conv_tile(tile)
        ops.FORCE_SPLITK = sk
        try:
            profiled()
            for _ in range(args.reps):
                rec = profiled()
                assert len(rec) == n
                for i, (name, desc, us) in enumerate(rec):
                    t = TILE_OF[name.rsplit("_", 1)[0]]   # strip the dtype suffix
                    s = int(re.search(r"sk(\d+)", desc).group(1))
                    results[i][(t, s)].append(us)
        finally:
            lib.madm_debug_set_conv_tile(0)
            ops.FORCE_SPLITK = None

    run(0, None, "current table")
    current = [min(((k, statistics.median(v)) for k, v in r.items()), key=lambda kv: kv[1]) for r in results]
    for tile in args.tiles:
        for sk in args.splitk:
            run(tile, sk, f"t{tile} sk{sk}")

    shapes = collections.OrderedDict()   # desc without the sk suffix -> list of launch indices
    for i, (name, desc, _) in enumerate(base):
        shapes.setdefault(re.sub(r" sk\d+$", "", desc), []).append(i)
    rows, tot_cur, tot_best = [], 0.0, 0.0
    for key, idxs in shapes.items():
        cand = collections.defaultdict(list)
        for i in idxs:
            for k, v in results[i].items():
                cand[k].append(statistics.median(v))
        # a candidate must have been measured on every launch of the shape
        full = {k: sum(v) for k, v in cand.items() if len(v) == len(idxs)}
        best = min(full.items(), key=lambda kv: kv[1])
        cur = sum(current[i][1] for i in idxs)
        cur_cfg = current[idxs[0]][0]
        tot_cur += cur
        tot_best += best[1]
        mm = re.match(r"M(\d+) N(\d+) K(\d+) k(\d+)", key)
        variant = 1 if " gn" in key else (2 if " up" in key else 0)
        rows.append((best[1], key, len(idxs), cur_cfg, cur / len(idxs), best[0], best[1] / len(idxs),
                     tuple(int(x) for x in mm.groups()) + (variant,)))
    rows.sort(reverse=True)
    print(f"current table {tot_cur / 1e3:.3f} ms -> best per shape {tot_best / 1e3:.3f} ms (HIP-event time, eager)")
    print(f"{'shape':42s} {'n':>3s} {'current':>10s} {'us':>8s} {'best':>10s} {'us':>8s}")
    for r in rows:
        print(f"{r[1]:42s} {r[2]:3d}  t{r[3][0]}/sk{r[3][1]:<3d} {r[4]:8.1f}   t{r[5][0]}/sk{r[5][1]:<3d} {r[6]:8.1f}")
    print("\n// ---- rows for igemm_tuned.inc: {dtype, M, N, K, KH, variant, tile, splitk}")
    seen = set()
    dt = 0 if args.dtype == "f32" else 1      # one table serves both 16-bit types
    for r in sorted(rows, key=lambda r: r[7]):
        if r[7] in seen:      # same GEMM shape and variant reached through a different stride / source layout
            continue
        seen.add(r[7])
        M, N, K, KH, variant = r[7]
        print(f"{{{dt}, {M}, {N}, {K}, {KH}, {variant}, {r[5][0]}, {r[5][1]}}},")


if __name__ == "__main__":
    main()
