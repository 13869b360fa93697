#!/usr/bin/env python3
"""bench.py -- UNet feature-extract images/sec @512x512, bs=2/GPU (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic images already resident in HBM:
VAE encode -> add_noise -> UNet (LoRA off, shipped configs) -> taps [1280@16, 640@32, 320@64] handed
over as NCHW f32, i.e. LdmRocm.forward (== LdmDiffusers.forward, ldm_diffusers.py:143-217).  Every
step goes through the product runner (madm_amd/pipeline.py::StagedExtractor.submit with a DIFFERENT
seeded batch: hipGraph executables of the two stages on four streams, the reference's input-range
assert kept as a deferred check); timing brackets exactly K submits with barrier + synchronize,
max over ranks.

Multi-GPU: the path shards by image with no exchange step ("replicas only"): every
rank runs its own bs=2 batch; RCCL is used only for the barrier and the max-over-ranks of the time.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time



def _early_int_flag(name, default):
    """Flags that must act before the HIP runtime starts (it reads its environment once)."""
    for i, a in enumerate(sys.argv):
        if a == name and i + 1 < len(sys.argv):
            return int(sys.argv[i + 1])
        if a.startswith(name + "="):
            return int(a.split("=", 1)[1])
    return default


# The staged pipeline below keeps 1 + --pipeline streams busy; ROCm multiplexes HIP streams onto GPU_MAX_HW_QUEUES
# hardware queues (default 4), and streams that share a queue serialise (measured: 4 UNet graphs side by side take
# 4.27 ms each with 4 queues, 3.47 ms with 8 -- profiles/round2_unet_concurrency.txt).
HW_QUEUES = _early_int_flag("--hw-queues", 8)
if HW_QUEUES > 0:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(HW_QUEUES))

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_FLOP_PER_IMAGE = 1.91993e12   # VAE-enc 1116.66 + UNet 803.27 GFLOP
PEAK_BF16_TFLOPS = 2500.0         # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level table)
PEAK_F32_TFLOPS = 157.3
IN_KERNEL_GHZ = 1.9          # d s_memtime / d s_memrealtime inside conv3x3_h16 under load (profiles/round6_h16_realtime.txt)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2, help="images per GPU (BASELINE: 2)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="f16", choices=["bf16", "f16", "f32"],
                    help="storage / MFMA input type (f32 accumulate).  f16 is the default: the reference's own autocast "
                         "arithmetic (engine/train_loop.py:277), same MFMA rate as bf16 (measured 269.5 vs 272.8 images/s, "
                         "same box) and 8x closer to the f32 oracle (rel. L2 1.7-3.3e-3 vs 1.3-2.7e-2 per tap, "
                         "profiles/round2_precision_f16_bf16.txt)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--graphs", type=int, default=1,
                    help="graph executables replayed round-robin (measured: 1, 2 and 3 give the same step time, the "
                         "host-side launch of a replay already overlaps the previous one)")
    ap.add_argument("--streams", type=int, default=4,
                    help="whole-forward mode (eval / slide workloads, or --pipeline 0): HIP streams the graph executables are "
                         "replayed on, round-robin, so consecutive steps overlap.  Four hardware pipes serve the queues: one stream "
                         "per pipe (round 5, same box, eval images/s on 2 / 3 / 4 / 5 / 6 / 8 streams: 103.7 / 111.6 / 116.8 / 97.7 / "
                         "103.9 / 104.2; sliding windows 3 / 4 / 5: 45.3 / 46.7 / 44.6; profiles/round5_ab_eval_streams.txt)")
    ap.add_argument("--pipeline", type=int, default=3,
                    help="extract workload: UNet streams of the staged pipeline (madm_amd/pipeline.py: every batch's VAE "
                         "encoder on one stream, its UNet on one of K streams, so K UNets of consecutive batches run side by "
                         "side and the next encoder slides under them; K + 1 batches in flight).  0 = whole-forward graphs "
                         "round-robin on --streams streams (the round-1 launch)")
    ap.add_argument("--slots", type=int, default=6,
                    help="extract workload: in-flight slots of the staged pipeline (0 = one per UNet stream).  More slots than "
                         "streams let the in-order encoder stream run ahead instead of waiting for the UNet of three submits ago: "
                         "same box, images/s at 3 / 4 / 5 / 6 slots: 358.7 / 348.8 / 365.8 / 363.1; a slower box at 3 / 5 / 7 / 8 / "
                         "9 / 11: 351.8 / 355.2 / 349.0 / 356.7 / 355.2 / 356.0 (profiles/round5_ab_slots.txt); 2 x streams keeps "
                         "the slot -> stream map balanced")
    ap.add_argument("--host-inputs", action="store_true",
                    help="extract workload, informational: the input pool lives in PINNED HOST memory and every submit transfers "
                         "its batch over PCIe (the contract's `value` is measured with the inputs resident in HBM; this is the "
                         "PCIe-inclusive rate DESIGN.md section 6 quotes)")
    ap.add_argument("--hw-queues", type=int, default=8,
                    help="GPU_MAX_HW_QUEUES for this process (0 = leave the runtime default of 4)")
    ap.add_argument("--eval-runner", default="graphed", choices=["graphed", "staged"],
                    help="eval / slide workloads: whole-forward graphs on --streams streams (GraphedInference) or three stage "
                         "graphs per image, encoder | UNet x --pipeline | decoder + head (StagedInference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--lora", action="store_true", help="enable one r=8 LoRA adapter (north_star variant)")
    ap.add_argument("--workload", default="extract", choices=["extract", "eval", "slide", "train"],
                    help="extract = BASELINE configs[1] (the metric); eval = configs[2]: full meta-arch inference "
                         "(VAE decoder + projections + DAFormer head, RGB->Depth config, batch 1); slide = configs[4] geometry: "
                         "the same on a 1024 x 512 image through sliding-window inference (three 512-wide windows as ONE "
                         "batched forward, feature_extractor.py:199-278, K = 9 Infrared head); train = configs[3]: one "
                         "MTMADISE training step (source + target + teacher pass, backward, clip, AdamW, EMA), RGB->Depth "
                         "config, full UNet fine-tune -- both informational")
    ap.add_argument("--color-aug", action="store_true", help="train workload: colour jitter + blur of strong_transform")
    ap.add_argument("--exchange", default="allreduce", choices=["allreduce", "rs_ag"],
                    help="train workload, N > 1: gradient exchange per bucket (dist.GradBucketReducer): all_reduce, or "
                         "reduce_scatter + all_gather (one direct exchange per peer on the xGMI mesh)")
    ap.add_argument("--wire", default="f32", choices=["f32", "bf16", "f16"], help="train workload: gradient wire type")
    ap.add_argument("--no-calib", action="store_true",
                    help="skip the two calibration measurements (calib / value_normalised in the line): for rocprofv3 passes, "
                         "whose kernel tables should hold the step's kernels only")
    ap.add_argument("--no-alt-dtype", action="store_true",
                    help="skip the second timed run of the extract workload in the other 16-bit type (alt_dtype in the line)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="--gpus N without a torchrun environment: spawn the N ranks, have each print its RANK / LOCAL_RANK / "
                         "WORLD_SIZE and exit before anything touches a GPU (launcher test)")
    return ap.parse_args()


def make_inputs(B, size, device, seed=0):
    img = torch.rand((B, 3, size, size), generator=torch.Generator().manual_seed(1234 + 7919 * seed))
    cond = 0.02 * torch.randn((1, 77, 768), generator=torch.Generator().manual_seed(1235 + 7919 * seed))
    return {"img": img.to(device), "cond_inputs": cond.repeat_interleave(B, 0).to(device),
            "cond_emb": torch.zeros((B, 1, 1280), device=device)}


POOL = 16   # distinct seeded batches resident in HBM; step i submits batch i mod POOL (a new batch on every step of the timed
# region at the driver's K = 20 up to the pool size, then the pool repeats: 16 x 6.3 MB of images)


def make_input_pool(B, size, device, n=POOL):
    return [make_inputs(B, size, device, seed=i) for i in range(n)]


def build_eval_model(dtype, device, finetune_unet='no', slide=False, num_classes=11):
    """BASELINE configs[2]: the shipped RGB->Depth inference graph (mtmadise_cityscapes_rgb_to_depth_11.py)."""
    from madm_amd.ldm_rocm import LdmRocm
    from madm_amd.backbone import BasePromptTimeGenerator, AttentionFeatureExtractorBackbone
    from madm_amd.head import DAFormerHead
    from madm_amd.meta_arch import MadmInference
    from madm_amd import weights
    ldm = LdmRocm("", encoder_block_indices=[], unet_block_indices=[5, 8, 11], decoder_block_indices=(),
                  input_range='-1+1', unet_block_indices_type='after', finetune_unet=finetune_unet, compute_dtype=dtype,
                  weights='synthetic', seed=0, vae_decoder_loss=True, device=device)
    gen = BasePromptTimeGenerator(learnable_cond_prompt=True, learnable_cond_time=True, clip_state='no', num_timesteps=1,
                                  ldm_extractor=ldm, same_cond_params=True)
    backbone = AttentionFeatureExtractorBackbone(
        attention_features_res=None, feature_dims=[3, 320, 640, 1280], projection_dim=[128, 512, 512, 512],
        attention_features_location=None, feature_extractor=gen, num_res_blocks=1, out_features=["s0", "s3", "s4", "s5"],
        slide_inference=slide)
    head = DAFormerHead(in_channels=[128, 512, 512, 512], in_keys=["s0", "s3", "s4", "s5"], in_index=[0, 1, 2, 3], channels=256,
                        dropout_ratio=0.1, num_classes=num_classes, norm_cfg=dict(type='BN'), align_corners=False,
                        decoder_params=dict(embed_dims=256, embed_cfg=dict(type='mlp'), embed_neck_cfg=dict(type='mlp'),
                                            fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False)))
    for prefix, mod in (("backbone.feature_projections.", backbone.feature_projections),
                        ("backbone.clip_project_rgb.", gen.clip_project_rgb), ("head.", head)):
        weights.synth_init_(mod, 0, prefix)
        weights.synth_buffers_(mod, 0, prefix)
    return MadmInference(backbone.to(device), head.to(device), target_modality="Depth").eval()


def build_train_model(dtype, device, color_aug):
    """BASELINE configs[3]: the shipped RGB->Depth training graph (mtmadise_cityscapes_rgb_to_depth_11.py: no LoRA,
    finetune_unet='all', vae_decoder_loss 'st' / L1, reg_uncertain, rev_noise_sup + gradually, timestep range [60, 61])."""
    from madm_amd.mtmadise import MTMADISE
    from madm_amd.criterion import CmdiseCriterion
    ev = build_eval_model(dtype, device, finetune_unet='all')
    palette = [int(v) for v in torch.randint(0, 256, (33,), generator=torch.Generator().manual_seed(99))]
    model = MTMADISE(ev.backbone, ev.sem_seg_head, CmdiseCriterion(num_classes=11), target_modality="Depth",
                     train_palette=palette, vae_decoder_loss='st', vae_decoder_loss_type='L1',
                     vae_decoder_loss_weight=[1.0, 1.0], reg_uncertain=True, rev_noise_sup=True, rev_noise_end_iter=5000,
                     rev_noise_gradually=True, denoise_timestep_range=[60, 61], max_iter=10000, color_aug_flag=color_aug)
    return model.train()


def train_inputs(B, size, device):
    g = torch.Generator().manual_seed(8899)
    out = []
    for _ in range(B):
        lab = torch.randint(0, 11, (1, size // 16, size // 16), generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2)
        lab[torch.rand((1, size, size), generator=g) < 0.06] = 255
        out.append({"source_rgb": (255.0 * torch.rand((3, size, size), generator=g)).to(device),
                    "source_label": lab.long().to(device),
                    "target_second_modality": (255.0 * torch.rand((3, size, size), generator=g)).to(device),
                    "width": size, "height": size})
    return out


def run_train(args, rank, world, device, dist, mdist):
    """configs[3]: K optimisation steps of MadmTrainer (eager launches; nothing is graph-captured yet)."""
    from madm_amd.train import MadmTrainer
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    model = build_train_model(dtype, device, args.color_aug)
    trainer = MadmTrainer(model, lr=5e-6, weight_decay=0.05, grad_clip=0.01, dist=dist, amp=True, exchange=args.exchange,
                          wire_dtype={"f32": None, "bf16": torch.bfloat16, "f16": torch.float16}[args.wire])
    data = train_inputs(args.batch, args.size, device)
    for _ in range(max(1, args.warmup)):
        losses, norm, stepped = trainer.run_step(data)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses, norm, stepped = trainer.run_step(data)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        elapsed = mdist.max_over_ranks(elapsed, dist, device)
    if rank == 0:
        n_param = sum(p.numel() for p in trainer.opt.params)
        out = {"metric": "MADM training step images/sec @512x512 bs=2/GPU (configs[3], informational)",
               "value": round(args.batch * world * args.steps / elapsed, 3), "unit": "images/s", "n_gpus": world,
               "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": round(1e3 * elapsed / args.steps, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "configs[3]: MTMADISE training step, RGB->Depth config (source + mixed-target gradient "
                                      "passes, EMA-teacher pass, 2 colour-label VAE encodes, CE + L1 losses, backward, "
                                      f"clip 0.01, AdamW, EMA), {args.batch}x3x{args.size}x{args.size} per GPU, full UNet fine-tune "
                                      f"({n_param / 1e6:.1f} M trainable parameters), colour augmentation "
                                      + ("on" if args.color_aug else "off"),
                          "global_batch": args.batch * world,
                          "parallelism": (f"dp{world}: gradient mean over the flat buffer, {args.exchange}, wire {args.wire}, "
                                          "256 MB buckets started during the backward") if world > 1 else "dp1",
                          "launch": "eager"},
               "allreduce_exposed_ms": trainer.last_allreduce_exposed_ms,
               "allreduce_started_during_backward_frac": trainer.last_overlap_frac,
               "last_losses": losses, "grad_norm": norm, "stepped": stepped, "loss_scale": trainer.scale,
               "peak_memory_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


def kernel_profile(model, inputs):
    """Two eager passes with HIP events around every MFMA-kernel launch (events recorded on the launch stream).

    * raw: one event pair per launch -- the bracket holds the kernel plus the marker / dispatch cost of the pair;
    * differenced (conv / GEMM / attention launches): the launch is issued three times as [K] [K K] between three events
      (ops.PROFILE_DIFF); bracket 2 - bracket 1 = one launch with the pair's constant cost cancelled.  This replaces
      round 2's "empty bracket" overhead estimate, which over-corrected (8.6 us subtracted per launch where
      rocprofv3's timestamps implied ~4.6).  Launches without a differenced figure (the stem) keep the raw one.
      (Round 4: attention too -- its 22 short launches at L <= 1024 read 18 .. 21 us each by single pairs against 8 .. 22 us by
      rocprofv3's timestamps of the same kernels, profiles/round4_final_last_replay.txt.)

    Returns ({kernel: (launches, raw_ms, algorithmic_flops, algorithmic_bytes, differenced_ms)},
             {kernel: {layer description "M.. N.. K.. ...": [launches, flops, bytes, differenced_ms]}})."""
    from madm_amd import ops
    # the table describes the kernels of the TIMED region: the runners capture under the throughput rows of the tile table, so the
    # eager profiling passes are pinned to them too (a plain forward() would take the lone-launch rows of the latency profile)
    with ops.tuning_profile("throughput", pin=True):
        for _ in range(2):
            model(*inputs)
    torch.cuda.synchronize()

    def one_pass(diff):
        ops.PROFILE, ops.PROFILE_DIFF = [], diff
        try:
            with ops.tuning_profile("throughput", pin=True):
                model(*inputs)
            torch.cuda.synchronize()
            return ops.PROFILE
        finally:
            ops.PROFILE, ops.PROFILE_DIFF = None, False

    raw = one_pass(False)
    # the differenced pass THREE times, per-launch median: one sample is at the mercy of a host hiccup between the second and
    # the third launch of a bracket (round 6, first bench process on a fresh box: one small launch read 60 x its time and
    # became the "dominant kernel" of the line)
    difs = [one_pass(True) for _ in range(3)]
    for dif in difs:
        assert [r[0] for r in raw] == [r[0] for r in dif]
    agg, layers = {}, {}
    for i, (name, flops, e0, e1, desc, nbytes, _p) in enumerate(raw):
        n, ms, fl, by, dms = agg.get(name, (0, 0.0, 0.0, 0, 0.0))
        t_raw = e0.elapsed_time(e1)
        samples = []
        for dif in difs:
            _, _, d0, d1, _, _, dp = dif[i]
            if dp.e2 is not None:
                samples.append(max(d1.elapsed_time(dp.e2) - d0.elapsed_time(d1), 1e-4))
        t_dif = sorted(samples)[len(samples) // 2] if samples else t_raw
        agg[name] = (n + 1, ms + t_raw, fl + flops, by + nbytes, dms + t_dif)
        row = layers.setdefault(name, {}).setdefault(desc, [0, 0.0, 0, 0.0])
        row[0] += 1
        row[1] += flops
        row[2] += nbytes
        row[3] += t_dif
    return agg, layers


CALIB_REF = {"mfma_loop_tflops": 2000.0, "h16_128x128_512sq_us": 141.0}   # the middle of the boxes seen in round 4 with
# exactly these two measurements (loop 1 835 .. 2 022 TFLOP/s, layer 139.5 .. 145.8 us): what value_normalised refers to


def calibrate(device, dtype):
    """Two fixed measurements of THIS device, outside every timed region (< 0.5 s): the chip-wide MFMA loop
    (madm_calib_mfma_loop: what its clock holds under matrix load) and one fixed layer of the dominant kernel (3x3 conv
    128 -> 128 channels on 2 x 512 x 512, tile 12, random data, cold-start excluded).  Boxes of the pool differ by +-7 .. 11 %
    on one tree (VERDICT r3 "What's weak" 14); these figures let driver lines of different rounds be compared."""
    import ctypes
    import math
    from madm_amd import ops
    from madm_amd._lib import lib
    sink = torch.zeros(1, device=device)
    flop = ctypes.c_double(0.0)
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 0.0
    for rep in range(3):
        e0.record()
        ops.check(lib.madm_calib_mfma_loop(60000, 512, sink.data_ptr(), ctypes.byref(flop), st), "madm_calib_mfma_loop")
        e1.record()
        torch.cuda.synchronize()
        best = max(best, flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    B, H, C = 2, 512, 128
    x = torch.randn((B * H * H, C), device=device).to(dtype)
    w = (torch.randn((C, 9 * C), device=device) / math.sqrt(9 * C)).to(dtype)
    bias = torch.randn(C, device=device)
    out = torch.empty((B * H * H, C), device=device, dtype=dtype)
    lib.madm_debug_set_conv_tile(12)
    try:
        for _ in range(3):
            ops.conv2d(x, w, B, H, H, N=C, KH=3, KW=3, pad_t=1, pad_l=1, bias=bias, out=out, splitk=1)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            ops.conv2d(x, w, B, H, H, N=C, KH=3, KW=3, pad_t=1, pad_l=1, bias=bias, out=out, splitk=1)
        e1.record()
        torch.cuda.synchronize()
    finally:
        lib.madm_debug_set_conv_tile(0)
    us = e0.elapsed_time(e1) * 1e3 / 10
    return {"mfma_loop_tflops": round(best, 1), "h16_128x128_512sq_us": round(us, 1),
            "reference": CALIB_REF,
            "note": "outside the timed region; value_normalised = value x (this box's h16 layer time / the reference box's): "
                    "what the same tree would read on the reference box if every kernel scaled like the dominant one"}


def pmc_traffic(kernel, workload):
    """Mean HBM bytes per launch of ``kernel`` from the committed PMC passes of this same command
    (tools/pmc.sh -> tools/pmc_report.py --json; counters cannot be read from inside the process)."""
    rel = os.path.join("profiles", f"pmc_bench_{workload}.json")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), rel)
    try:
        raw = open(path, "rb").read()
        import hashlib
        ent = json.loads(raw)["kernels"][kernel]
        return (ent["hbm_bytes_per_launch"],
                f"{rel}@sha256:{hashlib.sha256(raw).hexdigest()[:12]} (committed rocprofv3 --pmc passes of this command, "
                "tools/pmc.sh; counters cannot be read from inside the process)", ent.get("sq"))
    except (OSError, KeyError, ValueError):
        return None, None, None


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(size):
    """The CPU oracle (oracle/, a restatement -> kind "port") on the host cores, bounded: ONE image
    through the same VAE-enc -> add_noise -> UNet graph in fp32 (after a 64x64 warm-up that pages the
    3.8 GB of parameters in and spins the thread pool up).  Thread count = usable cores, at most 32
    (oneDNN convolutions on this graph stop scaling well before that)."""
    from oracle import sd_modules, ldm_path
    from madm_amd import weights
    threads = max(1, min(usable_cores(), 32))
    torch.set_num_threads(threads)
    vae = sd_modules.AutoencoderKL().eval()
    unet = sd_modules.UNet2DConditionModel().eval()
    weights.synth_init_(vae, 0, "vae.")
    weights.synth_init_(unet, 0, "unet.")
    sched = sd_modules.DDPMScheduler()
    cond = 0.02 * torch.randn((1, 77, 768), generator=torch.Generator().manual_seed(1235))
    noise = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(42))
    ts = torch.zeros(1, dtype=torch.int64)

    def run(sz):
        img = torch.rand((1, 3, sz, sz), generator=torch.Generator().manual_seed(1234))
        t0 = time.perf_counter()
        ldm_path.ldm_forward(vae, unet, sched, noise, img, cond, torch.zeros(1, 1, 1280), timesteps=ts)
        return time.perf_counter() - t0

    run(64)
    run(size)                                        # warm-up at the full size (1 warm-up + 3 timed, median)
    ts_ = sorted(run(size) for _ in range(3))
    t = ts_[1]
    return {"value": round(1.0 / t, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"1 image {size}x{size} x (1 warm-up + 3 timed, median; min {ts_[0]:.2f} max {ts_[2]:.2f} s), fp32 "
                      f"torch-CPU oracle on {threads} threads of {usable_cores()} usable cores: {t:.2f} s/image = "
                      f"{ALG_FLOP_PER_IMAGE / t / 1e9:.0f} GFLOP/s"}


def main():
    args = parse()
    from madm_amd import dist as mdist
    rank, local_rank, world = mdist.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = mdist.init("nccl", device)   # RCCL: barrier + max-over-ranks only, no data-path collective

    from madm_amd.ldm_rocm import LdmRocm
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    if args.workload == "train":
        return run_train(args, rank, world, device, dist, mdist)
    if args.workload in ("eval", "slide"):
        args.batch = 1
        slide = args.workload == "slide"
        model = build_eval_model(dtype, device, slide=slide, num_classes=9 if slide else 11)
        ldm = model.backbone.feature_extractor.ldm_extractor
        pool = [[{"target_second_modality": 255.0 * torch.rand((3, args.size, args.size * (2 if slide else 1)),
                                                                generator=torch.Generator().manual_seed(777 + i)).to(device)}]
                for i in range(POOL // 2)]
        return run(args, model, (pool[0],), ldm, rank, world, device, dist, mdist, pool)
    model = LdmRocm("", encoder_block_indices=[], unet_block_indices=[5, 8, 11], decoder_block_indices=[],
                    input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                    compute_dtype=dtype, weights='synthetic', seed=0, device=device)
    if args.lora:
        from types import SimpleNamespace
        from madm_amd import weights
        model.unet.add_adapter(SimpleNamespace(r=8, lora_alpha=8, target_modules=["to_k", "to_q", "to_v", "to_out.0"]),
                               "Depth")
        model.unet.set_adapter(["Depth"])
        weights.randomize_lora_B_(model.unet)
    pool = make_input_pool(args.batch, args.size, device)
    call = (pool[0], "rgb")
    if args.host_inputs:
        pool = [{k: v.cpu().pin_memory() for k, v in b.items()} for b in pool]
    return run(args, model, call, model, rank, world, device, dist, mdist, pool)


def run(args, model, call, ldm, rank, world, device, dist, mdist, pool):
    """``pool``: the distinct input batches of the workload (already in HBM); step i submits pool[i mod len(pool)]."""
    torch.set_grad_enabled(False)   # inference workloads (with --lora the adapters are trainable: no tape wanted here)
    # eager warm-up: packs weights, sizes workspaces; the reference's range assert (ldm_diffusers.py:147) runs as a host sync
    # here and in every other EAGER call below -- the graph runners keep it as a deferred check (pipeline.DeferredRangeCheck)
    model(*call)
    torch.cuda.synchronize()

    calib = None
    if rank == 0 and args.workload == "extract" and args.dtype in ("f16", "bf16") and not args.no_calib:
        # before the warm-up and the timed region: a kernel trace of this process ends with the step's own launches
        calib = calibrate(device, {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype])
    prof = layers = None
    if rank == 0 and not args.no_kernel_profile:
        prof, layers = kernel_profile(model, call)

    first_stream = None      # set when the steps run on streams of their own
    staged = args.workload == "extract" and args.pipeline > 0 and not args.no_graph
    graphed = args.workload in ("eval", "slide") and not args.no_graph
    graphs, outs, streams = [], [], [None]
    runner = None

    def capture_whole_forward(nstreams, nexec):
        """Whole-forward hipGraph executables over ONE fixed batch, round-robin on ``nstreams`` streams (reference points and
        profiling runs; the product runners are pipeline.StagedExtractor / GraphedInference)."""
        import contextlib
        from madm_amd import ops as _ops
        # several executables side by side = a throughput mode: pinned to the throughput rows of the tile table like the product
        # runners; ONE executable = the graph of a synchronous forward() (it opens the latency profile by itself)
        prof = _ops.tuning_profile("throughput", pin=True) if (nstreams > 1 or nexec > 1) else contextlib.nullcontext()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), prof:
            model(*call)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        sts = [torch.cuda.Stream() for _ in range(nstreams)] if nstreams >= 1 else [None]
        gs, os_ = [], []
        prof = _ops.tuning_profile("throughput", pin=True) if (nstreams > 1 or nexec > 1) else contextlib.nullcontext()
        with prof:
            for i in range(nexec):
                g = torch.cuda.CUDAGraph()
                st = sts[i % len(sts)]
                if st is not None:
                    st.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(st):
                        model(*call)          # sizes this stream's split-K workspace outside the capture
                    st.synchronize()
                    with torch.cuda.graph(g, stream=st):
                        os_.append(model(*call))
                else:
                    with torch.cuda.graph(g):
                        os_.append(model(*call))
                gs.append(g)
        return gs, os_, sts

    def serial_reference(g, st):
        # reference point, outside the timed region: the same K steps strictly one after the other (one executable,
        # one stream) -- the latency of a step; the timed region overlaps consecutive steps on several streams
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            for _ in range(3):
                g.replay()
            a.record()
            for _ in range(args.steps):
                g.replay()
            b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.steps

    if args.no_graph:
        def step(i):
            return model(*((pool[i % len(pool)],) + tuple(call[1:])))
    elif staged:
        # the pipeline's streams are the FIRST streams this process creates: four busy streams run side by side only when
        # they sit on four different hardware pipes, and the runtime hands queues out in creation order (measured:
        # encoder + 3 UNet streams 6.3 ms per step, the same with one more stream on the encoder's pipe 7.8 ms --
        # profiles/round2_unet_concurrency.txt); the whole-forward executable of the serial_* reference point is built
        # after the timed region
        from madm_amd.pipeline import StagedExtractor
        # A/B switches for the cost of the product API's per-submit work: MADM_EXP_NO_SYNC_INPUTS=1
        # drops the caller-stream event, MADM_EXP_NO_RANGE=1 the deferred range check
        runner = pipe = StagedExtractor(ldm, call[0], unet_streams=args.pipeline, slots=args.slots or None,
                                        sync_inputs=not int(os.environ.get("MADM_EXP_NO_SYNC_INPUTS", "0")),
                                        range_check=False if int(os.environ.get("MADM_EXP_NO_RANGE", "0")) else None)

        def step(i):
            return pipe.submit(pool[i % len(pool)])[0]
        first_stream = pipe.s_enc
    elif graphed:
        from madm_amd.pipeline import GraphedInference, StagedInference
        if args.eval_runner == "staged":
            runner = StagedInference(model, pool[0], unet_streams=max(1, min(args.pipeline, 2)), slots=args.slots or None)
        else:
            runner = GraphedInference(model, pool[0], streams=max(1, args.streams), slots=max(1, args.graphs, args.streams))

        def step(i):
            return runner.submit(pool[i % len(pool)])[0]
        first_stream = runner.streams_[0]
    else:
        graphs, outs, streams = capture_whole_forward(args.streams if args.streams > 1 else 0,
                                                      max(1, args.graphs, args.streams))

        def step(i):
            j = i % len(graphs)
            st = streams[j % len(streams)]
            if st is None:
                graphs[j].replay()
            else:
                with torch.cuda.stream(st):
                    graphs[j].replay()
            return outs[j]

        first_stream = streams[0]

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    serial_ms = None
    if graphed:
        serial_ms = None          # measured after the timed region on a whole-forward graph of a synchronous forward() (below)
    elif not args.no_graph and not staged and args.streams > 1:
        serial_ms = serial_reference(graphs[0], streams[0])
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # The timed region starts from an idle device (synchronize above) and ends with a device synchronize; `value` is the
    # wall clock between them.  With several streams NOTHING else is enqueued in the region -- no fork from / join into a
    # timing stream, no timing events: every variant of a device-side bracket measured in round 2 cost throughput (staged
    # pipeline, same box: 303 images/s bare, 274 with an idle timing stream that joins at the end, 217 forked from and
    # joined into the legacy null stream as in round 1; a queue that sits on a wait keeps its hardware pipe busy and the
    # pipe's other queue -- one of the four working streams -- starves).  device_ms_per_step exists for one stream only.
    # What a submit enqueues besides the two graph launches: an event record on the (idle) caller stream + the encoder
    # stream's wait for it, three device-to-device input copies, one 8-byte device-to-pinned copy of the range probe.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    single = first_stream is None
    t0 = time.perf_counter()
    if single:
        ev0.record()
    for i in range(args.steps):
        step(args.warmup + i)
    if single:
        ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        elapsed = mdist.max_over_ranks(elapsed, dist, device)
    range_checked = None
    if runner is not None:
        runner.drain()                # the range checks still pending (the device is idle: nothing to wait for)
        if runner.range_check is not None:
            range_checked = runner.range_check.checked
    alt = None
    conc = None
    if staged:
        conc = pipe.concurrency_probe()
        if world == 1 and args.dtype in ("f16", "bf16") and not args.no_alt_dtype and not args.lora:
            # configs[1] says bf16, the default arithmetic is f16 (the reference's autocast type): the other 16-bit type is
            # timed in the same process with the same launch strategy so that both are driver-visible
            # ... and f32 -- the exact-f32 MFMA mode (v_mfma_f32_16x16x4_f32, 1/16 of the 16-bit matrix rate): the ONE mode inside
            # north_star's 1e-3 relative fp32 tolerance (observed 2 .. 6e-6), so its throughput is on record too
            from madm_amd.ldm_rocm import LdmRocm
            alt = {}
            for other in ("bf16" if args.dtype == "f16" else "f16", "f32"):
                odt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[other]
                m2 = LdmRocm("", encoder_block_indices=[], unet_block_indices=[5, 8, 11], decoder_block_indices=[],
                             input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                             compute_dtype=odt, weights='synthetic', seed=0, device=device)
                m2(*call)
                torch.cuda.synchronize()
                pipe2 = StagedExtractor(m2, call[0], unet_streams=args.pipeline, streams=pipe.streams, slots=args.slots or None)
                steps2 = args.steps if other != "f32" else max(4, args.steps // 4)
                for i in range(args.warmup if other != "f32" else 2):
                    pipe2.submit(pool[i % len(pool)])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(steps2):
                    pipe2.submit(pool[(args.warmup + i) % len(pool)])
                torch.cuda.synchronize()
                el2 = time.perf_counter() - t1
                pipe2.drain()
                v2 = args.batch * steps2 / el2
                pk2 = PEAK_F32_TFLOPS if other == "f32" else PEAK_BF16_TFLOPS
                alt[other] = {"value": round(v2, 3), "unit": "images/s", "ms_per_step": round(1e3 * el2 / steps2, 4), "steps": steps2,
                              "whole_path_roofline_frac": round(v2 * ALG_FLOP_PER_IMAGE / (pk2 * 1e12), 4),
                              "peak_tflops": pk2,
                              "note": "same process, same staged pipeline, streams and inputs, run after the headline region"
                                      + ("; exact-f32 MFMA arithmetic: the mode that meets the 1e-3 relative fp32 tolerance of "
                                         "north_star (priced against the f32 matrix peak)" if other == "f32" else "")}
                del pipe2, m2
                torch.cuda.empty_cache()
        gs_, _, sts_ = capture_whole_forward(1, 1)
        serial_ms = serial_reference(gs_[0], sts_[0])
    elif graphed:
        # one image in flight: the whole-forward graph of a synchronous model(batched_inputs) call -- captured under the latency
        # profile of the tile table (ops.sync_profile), unlike the runner's graphs
        gs_, _, sts_ = capture_whole_forward(1, 1)
        serial_ms = serial_reference(gs_[0], sts_[0])

    if rank == 0:
        images = args.batch * world * args.steps
        value = images / elapsed
        peak = PEAK_BF16_TFLOPS if args.dtype in ("bf16", "f16") else PEAK_F32_TFLOPS
        # full eval forward 6 347.25 GFLOP per 512 x 512 image; sliding windows: 3 x (extractor + decoder +
        # projections) + the head on the 512 x 1024 canvas (2 x 1 821.27)
        alg = {"extract": ALG_FLOP_PER_IMAGE, "eval": 6.34725e12,
               "slide": 3 * (1.91993e12 + 2.51452e12 + 0.09153e12) + 2 * 1.82127e12}[args.workload]
        fed = (f"every step submits a different seeded batch (pool of {len(pool)}, "
               + ("in PINNED HOST memory: PCIe-inclusive, informational" if getattr(args, "host_inputs", False) else "resident in HBM")
               + ") through "
               "{}.submit(batched_inputs): the inputs are copied into the slot's static buffers on the device per step")
        if args.no_graph:
            launch = "eager launches, a different seeded batch per step"
        elif staged:
            launch = (f"staged hipGraph pipeline (madm_amd/pipeline.py): VAE-encoder graphs on 1 stream, UNet graphs on "
                      f"{args.pipeline} streams, {pipe.n_slots} slots, up to {pipe.n_slots + 1} batches ({(pipe.n_slots + 1) * args.batch} images) in "
                      f"flight, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}; "
                      + fed.format("StagedExtractor") + "; serial_* = whole-forward graph of a synchronous forward() (the "
                      "latency profile of the tile table: lone-launch rows), one batch in flight")
        elif graphed:
            launch = (f"three stage graphs per image (madm_amd/pipeline.py::StagedInference): VAE encoder | UNet on {runner.k} streams | "
                      f"VAE decoder + projections + head, {runner.n_slots} slots; " + fed.format("StagedInference")
                      ) if args.eval_runner == "staged" else (
                      f"whole-forward hipGraphs (madm_amd/pipeline.py::GraphedInference) on {max(1, args.streams)} streams: "
                      f"{max(1, args.streams)} images in flight; " + fed.format("GraphedInference")
                      + "; serial_* = one image in flight (whole-forward graph of a synchronous call: latency profile of the tile table)")
        else:
            launch = ("hipGraph replay of ONE captured batch, 1 stream: one batch in flight" if args.streams <= 1 else
                      f"hipGraph replay of ONE captured batch on {args.streams} streams: {args.streams} batches in flight; "
                      "serial_* = one batch in flight") + " (reference / profiling mode, not the product runner)"
        if runner is not None and runner.range_check is not None:
            rc = (f"deferred, on: the reference's per-call input-range assert (ldm_diffusers.py:147) is kept inside the timed "
                  f"region without its host sync -- every step's min / max probe is copied to pinned memory behind the encoder "
                  f"and asserted on the host by a later submit / drain ({range_checked} batches checked in this process)")
        elif args.no_graph:
            rc = "on, per call with its host sync, exactly as the reference (eager mode)"
        else:
            rc = "once before the timed region (fixed captured batch); its min / max probe kernel still runs every step"
        out = {
            "metric": {"extract": "UNet feature-extract images/sec @512x512 bs=2/GPU",
                       "eval": "full meta-arch eval images/sec @512x512 bs=1 (configs[2], informational)",
                       "slide": "sliding-window eval images/sec @1024x512 bs=1, 3 windows (configs[4] geometry, informational)"
                       }[args.workload],
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("configs[1]: SD-v1-4 VAE-encode + UNet single-timestep feature extractor "
                                    f"(taps 5,8,11 'after'), {args.batch}x3x{args.size}x{args.size} per GPU, t=0, "
                                    "LoRA " + ("r=8 on" if args.lora else "off (shipped configs)") +
                                    ", seeded synthetic weights") if args.workload == "extract" else
                       (("configs[2]: full MADM inference forward, RGB->Depth config (VAE enc -> UNet -> VAE dec -> "
                         f"GN projections -> DAFormer head @512x512, K=11), 1x3x{args.size}x{args.size} per GPU")
                        if args.workload == "eval" else
                        ("configs[4] geometry: sliding-window inference of a 1x3x512x1024 image (three 512-wide windows "
                         "batched as B=3 through VAE enc -> UNet -> VAE dec -> projections, window features averaged, "
                         "DAFormer head @512x1024, K=9)")),
                       "global_batch": args.batch * world, "parallelism": f"replicas x{world} (no collectives)",
                       "launch": launch, "range_check": rc},
            "device_ms_per_step": round(ev0.elapsed_time(ev1) / args.steps, 4) if single else None,
            "serial_ms_per_step": None if serial_ms is None else round(serial_ms, 4),
            "serial_images_per_s_per_gpu": None if serial_ms is None else round(args.batch / serial_ms * 1e3, 3),
            "whole_path_roofline_frac": round(value / world * alg / (peak * 1e12), 4),
        }
        if calib is not None:
            out["calib"] = calib
            out["value_normalised"] = round(value * calib["h16_128x128_512sq_us"] / CALIB_REF["h16_128x128_512sq_us"], 3)
        if alt is not None:
            out["alt_dtype"] = alt
        if conc is not None:
            out["config"]["pipeline_concurrency"] = {
                "unet_side_by_side_over_serial": round(conc, 3),
                "note": f"{args.pipeline} UNet graphs on their {args.pipeline} streams / ({args.pipeline} x one alone), probed after "
                        "the timed region; ~1.0 would mean the streams share a hardware pipe"}
        if prof:
            dom = max(prof.items(), key=lambda kv: kv[1][4])
            name, (n, ms_raw, fl, by, ms) = dom
            achieved = fl / (ms * 1e-3) / 1e12
            traffic, traffic_src, sq = pmc_traffic(name, args.workload)
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak,
                               "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                               "traffic": traffic, "traffic_source": traffic_src,
                               "algorithmic_bytes": by // n if by else None,
                               "launches_per_step": n, "kernel_ms_per_step": round(ms, 4),
                               "timing": "HIP events on the launch stream, differenced brackets [K] [K K] (no overhead "
                                         "estimate subtracted); frac_events_raw = one plain event pair per launch",
                               "kernel_ms_per_step_events_raw": round(ms_raw, 4),
                               "frac_events_raw": round(fl / (ms_raw * 1e-3) / 1e12 / peak, 4),
                               # the launches behind `achieved`, one row per distinct layer shape of the dominant kernel
                               # (M = output pixels of the batch, N = output channels, K = 9 x input channels)
                               "layers": [{"shape": d, "launches": r[0], "us_per_launch": round(1e3 * r[3] / r[0], 2),
                                           "tflops": round(r[1] / (r[3] * 1e-3) / 1e12, 1),
                                           "algorithmic_mb_per_launch": round(r[2] / r[0] / 1e6, 2)}
                                          for d, r in sorted(layers[name].items(), key=lambda kv: -kv[1][3])]}
            if sq and sq.get("kernel_us"):
                # do vector and matrix work co-execute in the dominant kernel?  From the same committed
                # PMC passes as `traffic` (p5 of tools/pmc.sh), mean over that kernel's dispatches; utilisations are per-SIMD busy
                # cycles over the dispatch's wall time at IN_KERNEL_GHZ (the clock the chip holds inside this kernel, measured
                # with s_memtime / s_memrealtime stamps: profiles/round6_h16_realtime.txt)
                cyc = sq["kernel_us"] * 1e3 * IN_KERNEL_GHZ
                out["roofline"]["pmc"] = {"mfma_util": round(sq["mfma_busy_cycles_per_simd"] / cyc, 3),
                                          "valu_active": round(sq["valu_active_cycles_per_simd"] / cyc, 3),
                                          "valu_mfma_coexec": round(sq["coexec_cycles_per_simd"] / cyc, 3),
                                          "coexec_share_of_valu": sq["coexec_share_of_valu"],
                                          "kernel_us_in_pass": sq["kernel_us"], "in_kernel_clock_ghz": IN_KERNEL_GHZ,
                                          "source": traffic_src}
            out["kernels"] = {k: {"launches": v[0], "ms": round(v[4], 4), "ms_events_raw": round(v[1], 4),
                                  "tflops": round(v[2] / (v[4] * 1e-3) / 1e12, 2) if v[4] > 0 else None}
                              for k, v in sorted(prof.items(), key=lambda kv: -kv[1][4])}
        if world == 1 and not args.no_cpu_baseline and args.workload == "extract":
            out["cpu_baseline"] = cpu_baseline(args.size)
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


_RESULT_FD = None


def emit(out):
    """The ONE JSON line of the contract, on the process's real stdout."""
    line = (json.dumps(out) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def launch_ranks():
    """``python bench.py --gpus N`` (N > 1) outside a torchrun environment: start the N ranks as child processes through
    ``torch.distributed.run`` BEFORE this process touches a GPU (a process that initialised the GPU must never exec; the
    parent only counts devices, which does not initialise it) and relay rank 0's single JSON line."""
    import socket
    import subprocess
    n = _early_int_flag("--gpus", 1)
    dry = "--dry-launch" in sys.argv
    visible = torch.cuda.device_count()
    if not dry and visible < n:
        raise SystemExit(f"bench.py: {n} GPUs requested, {visible} visible")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    lines = [l for l in proc.stdout.read().decode().splitlines() if l.startswith("{")]
    rc = proc.wait()
    for l in lines:
        print(l)
    sys.stdout.flush()
    raise SystemExit(rc if rc else (0 if lines else 1))


if __name__ == "__main__":
    if _early_int_flag("--gpus", 1) > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks()
    if "--dry-launch" in sys.argv:     # a spawned rank of the launcher test: report the environment, touch nothing
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
        sys.exit(0)
    # everything else that writes to file descriptor 1 (the RCCL version banner at process-group start / teardown, library
    # chatter) goes to stderr: rank 0's stdout carries exactly one line
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)
    main()
