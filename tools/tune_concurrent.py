#!/usr/bin/env python3
"""Tile / split-K choice per UNet layer shape under the LAUNCH STRATEGY of the metric (madm_amd/pipeline.py: three UNet
graphs side by side): every candidate is timed (a) alone on one stream and (b) as three graphs of the same layer on three
streams, cold (rotating) operands, and the table row is chosen by (b) -- the time a launch takes OUT OF THE CHIP when its
neighbours are other streams' launches, not its latency on an idle chip (tools/tune_insitu.py measures that).

The shapes come from one profiled eager forward (ops.PROFILE), so the list is the path's own.
    python tools/tune_concurrent.py [--dtype f16] [--streams 3] [--reps 12] [--min-us 60] [--rows out.txt]
"""
import argparse
import collections
import math
import os
import re
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LIN_TILES = [1, 2, 3, 7, 8, 11, 14, 15, 16, 17]
CONV_TILES = [2, 7, 8, 9, 10, 11, 12, 14, 15, 16, 17]
SPLITS = [1, 2, 3, 4, 6, 8, 12, 24]


def parse_desc(desc):
    m = re.match(r"M(\d+) N(\d+) K(\d+) k(\d+) s(\d+)( up)?( gn)?( ln)?( geglu)?( res)? sk(\d+)", desc)
    M, N, K, k, s, up, gn, ln, geglu, res, sk = m.groups()
    return dict(M=int(M), N=int(N), K=int(K), k=int(k), s=int(s), up=bool(up), gn=bool(gn), ln=bool(ln), geglu=bool(geglu),
                res=bool(res), sk=int(sk))


class Layer:
    """One launch of the shape with rotating operands; candidates are forced through the debug tile switch + splitk."""

    def __init__(self, d, B, dtype, rotate_bytes):
        from madm_amd import ops
        self.ops, self.d, self.B, self.dtype = ops, d, B, dtype
        k, M, N, K = d["k"], d["M"], d["N"], d["K"]
        self.Cin = K // (k * k)
        OH = int(round(math.sqrt(M // B)))
        assert B * OH * OH == M, d
        self.OH = OH
        self.IH = OH // 2 if d["up"] else OH * d["s"]
        wbytes = N * K * 2
        R = max(3, min(48, int(math.ceil(rotate_bytes / max(wbytes, 1)))))
        xbytes = B * self.IH * self.IH * self.Cin * 2
        Rx = max(2, min(8, int(math.ceil(64e6 / max(xbytes, 1)))))
        self.xs = [torch.randn((B * self.IH * self.IH, self.Cin), device="cuda").to(dtype) for _ in range(Rx)]
        self.ws = [(torch.randn((N, K), device="cuda") / math.sqrt(K)).to(dtype) for _ in range(R)]
        self.bias = torch.randn(N, device="cuda")
        self.gn = None
        if d["gn"]:
            sums = torch.zeros((B, self.Cin, 2), dtype=torch.float64, device="cuda")
            ops.groupnorm_stats(self.xs[0], B, self.IH * self.IH, sums)
            self.gn = ([sums], torch.rand(self.Cin, device="cuda") + 0.5, torch.randn(self.Cin, device="cuda") * 0.1, 32, 1e-5, True)
        self.st = torch.zeros((B, N, 2), dtype=torch.float64, device="cuda") if k == 3 else None
        # the layer's own epilogue: folded LayerNorm (row sums from the A fragments), GEGLU (half the columns stored, one erf
        # per pair), residual -- the tile that wins a plain GEMM of the shape need not win these
        self.ln = (torch.randn(N, device="cuda"), 1e-5) if d["ln"] else None
        self.epi = ops.EPI_GEGLU if d["geglu"] else ops.EPI_NONE
        ocols = N // 2 if d["geglu"] else N
        self.res = [torch.randn((M, ocols), device="cuda").to(dtype) for _ in range(Rx)] if d["res"] else None
        self.it = 0

    def run(self, sk):
        d = self.d
        self.it += 1
        x, w = self.xs[self.it % len(self.xs)], self.ws[self.it % len(self.ws)]
        k = d["k"]
        pad = k // 2
        return self.ops.conv2d(x, w, self.B, self.IH, self.IH, N=d["N"], KH=k, KW=k, stride=d["s"], pad_t=pad, pad_l=pad,
                               OH=self.OH, OW=self.OH, upsample=d["up"], bias=self.bias, stats=self.st, gn=self.gn, splitk=sk,
                               ln=self.ln, epilogue=self.epi,
                               residual=None if self.res is None else self.res[self.it % len(self.res)])


def time_candidate(layers, sk, reps, streams):
    """layers: one Layer per stream.  Returns (alone us / launch, side-by-side us / launch out of the chip)."""
    graphs = []
    for L, s in zip(layers, streams):
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            L.run(sk)
            L.run(sk)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                L.run(sk)
        graphs.append(g)
    torch.cuda.synchronize()
    import time

    def wall(js, n=3):
        best = 1e9
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j in js:
                with torch.cuda.stream(streams[j]):
                    graphs[j].replay()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best
    wall([0])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    alone = 1e9
    for _ in range(3):
        with torch.cuda.stream(streams[0]):
            e0.record(streams[0])
            graphs[0].replay()
            e1.record(streams[0])
        torch.cuda.synchronize()
        alone = min(alone, e0.elapsed_time(e1) * 1e3 / reps)
    wall(range(len(streams)))
    side = wall(range(len(streams)), n=4) * 1e6 / (reps * len(streams))
    return alone, side


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--min-us", type=float, default=50.0, help="skip shape classes whose launches sum to less (eager events)")
    ap.add_argument("--max-m", type=int, default=8192, help="largest M tuned (the VAE's 512^2 .. 128^2 maps lie above)")
    ap.add_argument("--rows", default="", help="write MADM_TUNED_FILE rows of the side-by-side winners here")
    ap.add_argument("--alone-rows", default="", help="write the LONE-launch winners here (rows of the latency profile: "
                    "madm_amd/csrc/igemm_tuned_latency.inc via tools/apply_alone_rows.py)")
    ap.add_argument("--only", default="", help="regex on the shape description")
    ap.add_argument("--lora", action="store_true", help="extract workload with one r = 8 adapter active (bench.py --lora): the "
                    "K-extended projection GEMMs and the skinny x A^T GEMMs get rows of their own")
    ap.add_argument("--workload", default="extract", choices=["extract", "eval", "slide", "train"],
                    help="train: the launches of one optimisation step (forward + data gradients; eager, one stream: use --streams 1); "
                         "rows are written only for shapes the table does not hold yet")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    from madm_amd.ldm_rocm import LdmRocm
    from madm_amd import ops
    from madm_amd._lib import lib
    import bench
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    if args.workload == "eval":      # BASELINE configs[2]: whole meta-arch inference forward, one image
        args.batch = 1
        m = bench.build_eval_model(dtype, torch.device("cuda"))
        call = ([{"target_second_modality": 255.0 * torch.rand((3, 512, 512)).cuda()}],)
    elif args.workload == "slide":   # configs[4] geometry: 512 x 1024 image, three 512-wide windows batched as B = 3 (square maps only:
        args.batch = 3               # the head's 512 x 1024 canvas is skipped by the Layer constructor)
        m = bench.build_eval_model(dtype, torch.device("cuda"), slide=True, num_classes=9)
        call = ([{"target_second_modality": 255.0 * torch.rand((3, 512, 1024)).cuda()}],)
    elif args.workload == "train":   # BASELINE configs[3]: one MadmTrainer step (its conv2d launches incl. the data gradients)
        from madm_amd.train import MadmTrainer
        torch.set_grad_enabled(True)
        model = bench.build_train_model(dtype, torch.device("cuda"), False)
        trainer = MadmTrainer(model, lr=5e-6, weight_decay=0.05, grad_clip=0.01, dist=None, amp=True)
        data = bench.train_inputs(args.batch, 512, torch.device("cuda"))
        m = lambda d: trainer.run_step(d)
        call = (data,)
    else:
        m = LdmRocm("", [], [5, 8, 11], [], input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                    compute_dtype=dtype, weights='synthetic', seed=0)
        call = (bench.make_inputs(args.batch, 512, torch.device("cuda")), "rgb")
        if args.lora:
            from types import SimpleNamespace
            from madm_amd import weights
            m.unet.add_adapter(SimpleNamespace(r=8, lora_alpha=8, target_modules=["to_k", "to_q", "to_v", "to_out.0"]), "Depth")
            m.unet.set_adapter(["Depth"])
            weights.randomize_lora_B_(m.unet)
    m(*call)
    torch.cuda.synchronize()
    ops.PROFILE = []
    m(*call)
    torch.cuda.synchronize()
    rec, ops.PROFILE = ops.PROFILE, None
    del m
    if args.workload == "train":
        del trainer, model
        torch.set_grad_enabled(False)
    torch.cuda.empty_cache()
    classes = collections.OrderedDict()
    for name, _, e0, e1, desc, _, _ in rec:
        if name.startswith(("attn", "stem_", "conv2d_wgrad")):
            continue
        key = re.sub(r" sk\d+( \+gn)?$", "", desc)
        c = classes.setdefault(key, {"n": 0, "us": 0.0, "name": name, "desc": desc})
        c["n"] += 1
        c["us"] += e0.elapsed_time(e1) * 1e3
    table_keys = set()
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "madm_amd", "csrc", "igemm_tuned.inc")
    for line in open(inc):
        mm = re.match(r"^\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\}", line)
        if mm:
            table_keys.add(tuple(int(mm.group(i)) for i in (2, 3, 4, 5, 6)))
    streams = [torch.cuda.Stream() for _ in range(args.streams)]
    rows, arows, tot = [], [], collections.defaultdict(float)
    print(f"{'shape':44s} {'n':>3s} | {'current':>9s} {'alone':>7s} {'side':>7s} | {'best alone':>10s} {'us':>7s} | {'best side':>10s} {'alone':>7s} {'side':>7s}")
    for key, c in sorted(classes.items(), key=lambda kv: -kv[1]["us"]):
        d = parse_desc(c["desc"])
        # the VAE's big maps fill the chip on their own: nothing to choose there
        if c["us"] < args.min_us or d["M"] > args.max_m or (args.only and not re.search(args.only, key)):
            continue
        try:
            layers = [Layer(d, args.batch, dtype, 300e6) for _ in range(args.streams)]
        except AssertionError:
            continue
        nk = d["K"] // 64
        tiles = LIN_TILES if d["k"] == 1 else CONV_TILES
        if d["gn"]:
            tiles = [9, 10, 12]
        res = {}
        lib.madm_debug_set_conv_tile(0)
        # graphs long enough (~1.5 ms) that the host's launch + synchronise (~30 us) does not decide the figure
        reps = max(args.reps, min(240, int(1500.0 / max(c["us"] / c["n"], 1.0))))
        cur = time_candidate(layers, None, reps, streams)
        for t in tiles:
            if t == 12 and (layers[0].OH < 16 or d["N"] < 128 or d["s"] != 1):
                continue
            if t in (9, 10) and (d["k"] != 3 or d["s"] != 1 or d["up"]):
                continue
            for sk in SPLITS:
                if sk > max(1, nk // 2) or (t == 13 and sk > 1) or (d["ln"] and sk > 1):
                    continue
                if t in (9, 10, 12) and sk > max(1, layers[0].Cin // 64):
                    continue
                lib.madm_debug_set_conv_tile(t)
                try:
                    res[(t, sk)] = time_candidate(layers, sk, reps, streams)
                except Exception as e:   # ineligible combination
                    print("   skip", key, t, sk, str(e)[:80])
                finally:
                    lib.madm_debug_set_conv_tile(0)
        if not res:
            continue
        ba = min(res.items(), key=lambda kv: kv[1][0])
        bs = min(res.items(), key=lambda kv: kv[1][1])
        n = c["n"]
        tot["cur_alone"] += n * cur[0]; tot["cur_side"] += n * cur[1]
        tot["ba_alone"] += n * ba[1][0]; tot["ba_side"] += n * ba[1][1]
        tot["bs_alone"] += n * bs[1][0]; tot["bs_side"] += n * bs[1][1]
        print(f"{key:44s} {n:3d} | sk{d['sk']:<7d} {cur[0]:7.1f} {cur[1]:7.1f} | t{ba[0][0]}/sk{ba[0][1]:<6d} {ba[1][0]:7.1f} | "
              f"t{bs[0][0]}/sk{bs[0][1]:<6d} {bs[1][0]:7.1f} {bs[1][1]:7.1f}", flush=True)
        variant = 1 if d["gn"] else (2 if d["up"] else (3 if d["s"] == 2 else 0))
        if ba[1][0] < 0.97 * cur[0]:      # (latency profile: every workload's shapes, the extractor's included)
            arows.append(f"1 {d['M']} {d['N']} {d['K']} {d['k']} {variant} {ba[0][0]} {ba[0][1]}   # alone {cur[0]:.1f} -> {ba[1][0]:.1f} us x {n} launches, side {cur[1]:.1f} -> {ba[1][1]:.1f}")
        if args.workload == "train" and (d["M"], d["N"], d["K"], d["k"], variant) in table_keys:
            continue   # (a shape of the extractor's table: tuned side by side, not to be replaced by a lone-launch choice)
        if bs[1][1] < 0.97 * cur[1]:
            rows.append(f"1 {d['M']} {d['N']} {d['K']} {d['k']} {variant} {bs[0][0]} {bs[0][1]}   # side {cur[1]:.1f} -> {bs[1][1]:.1f} us, alone {cur[0]:.1f} -> {bs[1][0]:.1f}")
        del layers
        torch.cuda.empty_cache()
    print("sums over the forward's launches (us):", {k: round(v, 1) for k, v in tot.items()})
    if args.rows:
        with open(args.rows, "w") as f:
            f.write("# dtype M N K KH variant tile splitk -- tools/tune_concurrent.py, side-by-side winners\n")
            f.write("\n".join(rows) + "\n")
    print("\n".join(rows))
    if args.alone_rows:
        with open(args.alone_rows, "w") as f:
            f.write("# dtype M N K KH variant tile splitk -- tools/tune_concurrent.py, lone-launch winners (latency profile)\n")
            f.write("\n".join(arows) + "\n")
    print("lone-launch rows:")
    print("\n".join(arows))


if __name__ == "__main__":
    main()
