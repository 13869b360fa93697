"""How much do k UNet forwards gain from running side by side?  (bs = 2 UNet at 64 x 64 latents is latency-bound.)

Captures the VAE-encoder stage and the UNet stage of LdmRocm.forward as separate hipGraph executables
(`LdmRocm._stage_encode / _stage_unet`) and times
  (a) the encoder graph alone, back to back on one stream,
  (b) k UNet graphs replayed on k streams, k = 1 .. 4,
  (c) the staged pipeline: all encoders on one stream, UNet of step i on stream 1 + i % k after its encoder's event.

usage: python tools/exp/unet_concurrency.py [--dtype f16] [--rounds 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from madm_amd import ops  # noqa: E402
from madm_amd.ldm_rocm import LdmRocm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--kmax", type=int, default=4)
    ap.add_argument("--dummy", type=int, default=0, help="idle streams created (and used once) before the pipeline's")
    ap.add_argument("--only", default="", help="comma list of k: only the single-encoder-stream pipeline for these k")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    model = LdmRocm("", encoder_block_indices=[], unet_block_indices=[5, 8, 11], decoder_block_indices=[],
                    input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                    compute_dtype=dtype, weights='synthetic', seed=0, device=dev)
    inputs = bench.make_inputs(2, 512, dev)
    with torch.no_grad():
        model(inputs, "rgb")
        torch.cuda.synchronize()
        model.check_input_range = False
        K = args.kmax
        dummies = [torch.cuda.Stream() for _ in range(args.dummy)]
        for d in dummies:
            with torch.cuda.stream(d):
                torch.zeros(4, device=dev)
        torch.cuda.synchronize()
        s_enc = torch.cuda.Stream()
        s_un = [torch.cuda.Stream() for _ in range(K)]

        # encoder graphs: one per hand-over slot
        enc_graphs, slots = [], []
        for i in range(K):
            s_enc.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_enc):
                model._stage_encode(inputs)
            s_enc.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s_enc):
                st = model._stage_encode(inputs)
            enc_graphs.append(g)
            slots.append(st)
        un_graphs, outs = [], []
        for i in range(K):
            st_ = s_un[i]
            st_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st_):
                ops.ARENA.reset(dev)
                model._stage_unet(slots[i], inputs)
            st_.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st_):
                ops.ARENA.reset(dev)
                outs.append(model._stage_unet(slots[i], inputs))
            un_graphs.append(g)
        torch.cuda.synchronize()

        def timed(fn, n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / n

        R = args.rounds

        def enc_only():
            with torch.cuda.stream(s_enc):
                for r in range(R):
                    enc_graphs[r % K].replay()
        enc_only()
        t_enc = timed(enc_only, R)
        print(f"encoder stage alone: {t_enc:.3f} ms per batch", flush=True)

        only = [int(x) for x in args.only.split(",") if x]
        for k in ([] if only else range(1, K + 1)):
            def un_k():
                for r in range(R):
                    for j in range(k):
                        with torch.cuda.stream(s_un[j]):
                            un_graphs[j].replay()
            un_k()
            t = timed(un_k, R * k)
            print(f"{k} UNet graphs side by side: {t:.3f} ms per UNet  -> step {t_enc + t:.3f} ms = "
                  f"{2e3 / (t_enc + t):.1f} images/s if the encoder runs exclusively", flush=True)

        # staged pipeline
        for k in (only or range(1, K + 1)):
            evs = [torch.cuda.Event() for _ in range(K)]
            done = [torch.cuda.Event() for _ in range(K)]

            def pipe():
                for r in range(R):
                    j = r % k
                    with torch.cuda.stream(s_enc):
                        if r >= k:
                            s_enc.wait_event(done[j])     # slot j is free again
                        enc_graphs[j].replay()
                        evs[j].record(s_enc)
                    with torch.cuda.stream(s_un[j]):
                        s_un[j].wait_event(evs[j])
                        un_graphs[j].replay()
                        done[j].record(s_un[j])
            pipe()
            t = timed(pipe, R)
            print(f"pipeline, encoder stream + {k} UNet streams: {t:.3f} ms per step = {2e3 / t:.1f} images/s", flush=True)
        if only:
            return

        # two encoder streams + k UNet streams
        s_enc2 = torch.cuda.Stream()
        for k in range(2, K + 1):
            evs = [torch.cuda.Event() for _ in range(K)]
            done = [torch.cuda.Event() for _ in range(K)]

            def pipe2():
                for r in range(R):
                    j = r % k
                    se = s_enc if (r & 1) == 0 else s_enc2
                    with torch.cuda.stream(se):
                        if r >= k:
                            se.wait_event(done[j])
                        enc_graphs[j].replay()
                        evs[j].record(se)
                    with torch.cuda.stream(s_un[j]):
                        s_un[j].wait_event(evs[j])
                        un_graphs[j].replay()
                        done[j].record(s_un[j])
            pipe2()
            t = timed(pipe2, R)
            print(f"pipeline, 2 encoder streams + {k} UNet streams: {t:.3f} ms per step = {2e3 / t:.1f} images/s", flush=True)

        # batched-phase schedule: k encoders back to back, then k UNets side by side, phases separated by events
        for k in range(2, K + 1):
            def phased():
                main_s = s_enc
                for r in range(R // k):
                    with torch.cuda.stream(main_s):
                        for j in range(k):
                            main_s.wait_stream(s_un[j])
                        for j in range(k):
                            enc_graphs[j].replay()
                    for j in range(k):
                        with torch.cuda.stream(s_un[j]):
                            s_un[j].wait_stream(main_s)
                            un_graphs[j].replay()
            phased()
            t = timed(phased, (R // k) * k)
            print(f"phased, {k} encoders then {k} UNets side by side: {t:.3f} ms per step = {2e3 / t:.1f} images/s",
                  flush=True)


if __name__ == "__main__":
    main()
