"""Throughput launch strategy for LdmRocm.forward: the two stages of the path as separate hipGraph executables on
separate HIP streams.

Stage 1 (``LdmRocm._stage_encode``: normalise -> vae_encoder -> timestep draw -> add_noise, ldm_diffusers.py:143-163) is
MFMA-bound and fills the chip on its own; stage 2 (``_stage_unet``: diffusion_unet + the tap hand-over, :165-217) at
bs = 2 is launch / latency-bound and leaves most CUs idle.  So consecutive batches are overlapped
like this: every batch's encoder runs on ONE stream (encoders never overlap each other: nothing to gain), its UNet on
one of ``unet_streams`` streams after the encoder's event, so up to ``unet_streams`` UNets of different batches run side
by side and the next encoder slides under them.  Each in-flight batch owns a hand-over slot (latents, timesteps) and its
own output tensors; a slot's encoder waits until the slot's previous UNet has finished.

Same kernels, same per-batch results as ``LdmRocm.forward`` (tests/test_parity_gpu.py::test_staged_pipeline_matches_forward);
only the order in which the GPU sees the launches changes.
"""
import os
import warnings

import torch

from . import ops


class StagedExtractor:
    def __init__(self, ldm, batched_inputs, unet_streams=3, streams=None, **kwargs):
        assert unet_streams >= 1
        self.ldm = ldm
        self.k = int(unet_streams)
        # the k + 1 streams overlap only when each sits on a hardware queue (and pipe) of its own: the runtime multiplexes
        # HIP streams onto GPU_MAX_HW_QUEUES queues (default 4) in creation order and reads the variable once, at start-up
        q = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)
        if q < self.k + 1:
            warnings.warn(f"StagedExtractor: {self.k + 1} streams on GPU_MAX_HW_QUEUES={q} hardware queues -- streams will "
                          "share queues and serialise; export GPU_MAX_HW_QUEUES>=%d before the process starts" % (self.k + 1))
        dev = batched_inputs['img'].device
        if streams is not None:       # reuse another (idle) pipeline's streams: they already sit on pipes of their own
            assert len(streams) == self.k + 1
            self.s_enc, self.s_unet = streams[0], list(streams[1:])
        elif os.environ.get("MADM_EXP_CUMASK"):
            # experiment: "<enc_lo>-<enc_hi>,<unet_lo>-<unet_hi>" = CU ranges (of 256) the encoder stream / the UNet streams may
            # use (hipExtStreamCreateWithCUMask): does keeping the chip-filling encoder kernels off some CUs shorten the wait
            # of the UNet's small grids for slots?  
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")

            def masked(lo, hi):
                words = (ctypes.c_uint32 * 8)()
                for b_ in range(lo, hi):
                    words[b_ // 32] |= 1 << (b_ % 32)
                st = ctypes.c_void_p()
                rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
                assert rc == 0, f"hipExtStreamCreateWithCUMask failed: {rc}"
                return torch.cuda.ExternalStream(st.value, device=dev)
            enc, un = os.environ["MADM_EXP_CUMASK"].split(",")
            e0, e1 = (int(v) for v in enc.split("-"))
            u0, u1 = (int(v) for v in un.split("-"))
            self.s_enc = masked(e0, e1)
            self.s_unet = [masked(u0, u1) for _ in range(self.k)]
        else:
            # experiment switch MADM_EXP_PRIO: "u" = UNet streams high priority, "e" = encoder stream high priority
            prio = os.environ.get("MADM_EXP_PRIO", "")
            self.s_enc = torch.cuda.Stream(device=dev, priority=-1 if prio == "e" else 0)
            self.s_unet = [torch.cuda.Stream(device=dev, priority=-1 if prio == "u" else 0) for _ in range(self.k)]
        self.enc_graphs, self.unet_graphs, self.slots, self.outs = [], [], [], []
        cur = torch.cuda.current_stream(dev)
        with torch.no_grad():
            for j in range(self.k):
                self.s_enc.wait_stream(cur)
                with torch.cuda.stream(self.s_enc):
                    ldm._stage_encode(batched_inputs)          # sizes this stream's workspaces outside the capture
                self.s_enc.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.s_enc):
                    st = ldm._stage_encode(batched_inputs)
                self.enc_graphs.append(g)
                self.slots.append(st)
            for j in range(self.k):
                s = self.s_unet[j]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    ops.ARENA.reset(dev)
                    ldm._stage_unet(self.slots[j], batched_inputs, **kwargs)
                s.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    ops.ARENA.reset(dev)                       # the statistics arena of this stage: zeroed inside the graph
                    self.outs.append(ldm._stage_unet(self.slots[j], batched_inputs, **kwargs))
                self.unet_graphs.append(g)
        torch.cuda.synchronize(dev)
        self.encoded = [torch.cuda.Event() for _ in range(self.k)]
        self.done = [torch.cuda.Event() for _ in range(self.k)]
        self.turn = 0

    @property
    def streams(self):
        return [self.s_enc, *self.s_unet]

    def submit(self):
        """Enqueues one batch; returns (outputs, event): the slot's output tensors are valid once ``event`` has fired and
        stay so until the slot comes round again (``unet_streams`` submits later)."""
        j = self.turn % self.k
        first = self.turn < self.k
        self.turn += 1
        with torch.cuda.stream(self.s_enc):
            if not first:
                self.s_enc.wait_event(self.done[j])            # the slot's hand-over buffers are free again
            self.enc_graphs[j].replay()
            self.encoded[j].record(self.s_enc)
        s = self.s_unet[j]
        with torch.cuda.stream(s):
            s.wait_event(self.encoded[j])
            self.unet_graphs[j].replay()
            self.done[j].record(s)
        return self.outs[j], self.done[j]

    def concurrency_probe(self, reps=3):
        """Detects streams that share a hardware pipe: time of the k UNet graphs side by side on their k streams over k
        times the time of one of them alone (1.0 = fully serialised, ~0.6 measured for k = 3 on four free pipes: 3.7 vs
        5.7 ms per UNet).  Wall clock around device synchronisations; call it outside timed regions."""
        import time
        dev = self.s_enc.device

        def timed(js):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reps):
                for j in js:
                    with torch.cuda.stream(self.s_unet[j]):
                        self.unet_graphs[j].replay()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / reps

        timed(range(self.k))
        one = min(timed([j]) for j in range(self.k))
        side = timed(range(self.k))
        return side / (self.k * one)

    def fork(self, stream=None):
        """The pipeline's streams start after everything queued on ``stream`` (default: the current one)."""
        stream = stream or torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(stream)

    def join(self, stream=None):
        stream = stream or torch.cuda.current_stream()
        for s in self.streams:
            stream.wait_stream(s)
