"""Throughput launch strategy for LdmRocm.forward / MadmInference.forward: hipGraph executables on several HIP streams,
fed with a NEW batch per submit the way the reference's inference loop feeds its model
(/root/reference/evaluation/evaluator.py:75-93: ``for idx, inputs in enumerate(data_loader): outputs = model(inputs)``).

``StagedExtractor`` -- the two stages of the extractor as separate graphs on separate streams.
Stage 1 (``LdmRocm._stage_encode``: normalise -> vae_encoder -> timestep draw -> add_noise, ldm_diffusers.py:143-163) is
MFMA-bound and fills the chip on its own; stage 2 (``_stage_unet``: diffusion_unet + the tap hand-over, :165-217) at
bs = 2 is launch / latency-bound and leaves most CUs idle.  So consecutive batches are overlapped
like this: every batch's encoder runs on ONE stream (encoders never overlap each other: nothing to gain), its UNet on
one of ``unet_streams`` streams after the encoder's event, so up to ``unet_streams`` UNets of different batches run side
by side and the next encoder slides under them.  Each in-flight batch owns a SLOT: static input buffers (``img``,
``cond_inputs``, ``cond_emb``: ``submit`` copies the caller's tensors into them on the encoder stream), the hand-over
(latents, timesteps) and its own output tensors; a slot's copies and encoder wait until the slot's previous UNet has finished.

``GraphedInference`` -- the whole ``MadmInference.forward`` (pad -> prompt -> extractor -> VAE decoder -> projections ->
DAFormer head -> resize) as one graph per slot, slots round-robin on ``streams`` streams (default 4: one per hardware pipe of
the queue scheduler; a fifth stream shares a pipe and loses 15 %), same submit contract.

The reference's per-call input-range assert (ldm_diffusers.py:147: ``assert -1 <= images.min() and images.max() <= 1``, a host
sync per call) is kept as a DEFERRED check: the stem kernel's min / max probe of every submitted batch is copied to a pinned
ring behind the encoder and compared on the host as soon as a later ``submit`` / ``drain`` finds it completed -- the
``AssertionError`` comes a few submits late instead of stalling the device once per batch.

Same kernels, same per-batch results as ``LdmRocm.forward`` (tests/test_parity_gpu.py::test_staged_pipeline_*: distinct batches
per submit, every slot bit-identical to ``forward()`` on ITS batch); only the order in which the GPU sees the launches changes.
"""
import os
import warnings

import torch

from . import ops


class DeferredRangeCheck:
    """The reference's ``assert -1 <= images.min() and images.max() <= 1`` (ldm_diffusers.py:147) without its host sync: ring
    of pinned (min, max) pairs + one event per entry; ``push`` enqueues the device -> pinned copy on the given stream,
    ``poll`` checks every entry whose event has fired (non-blocking), ``drain`` waits for all of them."""

    def __init__(self, device, depth=64):
        self.depth = int(depth)
        self.host = torch.empty((self.depth, 2), dtype=torch.float32).pin_memory()
        self.events = [torch.cuda.Event() for _ in range(self.depth)]
        self.turn_of = [None] * self.depth
        self.head = 0            # next entry to check
        self.tail = 0            # next entry to fill
        self.checked = 0

    def make_room(self):
        """Called at the TOP of a submit, before anything of it is enqueued: polls, and if the ring is full waits for the oldest
        entry (`depth` submits old).  An AssertionError of an earlier batch therefore never leaves a half-enqueued submit
        behind -- ``push`` itself cannot raise."""
        self.poll()
        while self.tail - self.head >= self.depth:
            self._check(self.head, block=True)

    def push(self, minmax, stream, turn):
        assert self.tail - self.head < self.depth, "DeferredRangeCheck.push without make_room()"
        r = self.tail % self.depth
        self.host[r].copy_(minmax, non_blocking=True)
        self.events[r].record(stream)
        self.turn_of[r] = turn
        self.tail += 1

    def _check(self, i, block):
        r = i % self.depth
        if block:
            self.events[r].synchronize()
        elif not self.events[r].query():
            return False
        lo, hi = self.host[r].tolist()
        self.head = i + 1
        self.checked += 1
        assert -1 <= lo and hi <= 1, (f"input range check (ldm_diffusers.py:147), deferred: the batch of submit #{self.turn_of[r]} "
                                      f"has min {lo} max {hi} after normalisation, outside [-1, 1]")
        return True

    def poll(self):
        while self.head < self.tail and self._check(self.head, block=False):
            pass

    def drain(self):
        while self.head < self.tail:
            self._check(self.head, block=True)

    def discard(self):
        """Forgets the pending entries (their events must have fired: call after the streams were synchronised)."""
        self.head = self.tail


class Submitted(tuple):
    """What ``submit`` returns: unpacks as ``(outputs, done)`` (``GraphedInference``: ``(outputs, done, slot)``);
    ``.taken`` is the event behind the pipeline's copies of the caller's input tensors -- the caller may OVERWRITE those
    tensors once it has fired (``stream.wait_event(r.taken)`` / ``r.taken.synchronize()``), exactly as the source of any
    asynchronous copy; merely dropping them is safe at once (the copies are registered with the caching allocator);
    ``.turn`` numbers the submit (the deferred range assert names it)."""

    def __new__(cls, items, taken, turn):
        self = super().__new__(cls, items)
        self.taken, self.turn = taken, turn
        return self


def _static_like(t):
    return None if t is None else torch.empty_like(t, memory_format=torch.contiguous_format).copy_(t)


def _copy_checked(dst, src, name, stream):
    if dst is None:
        assert src is None, f"{name}: the pipeline was built without this input"
        return
    assert src is not None and tuple(src.shape) == tuple(dst.shape), \
        f"{name}: the graphs were captured for shape {tuple(dst.shape)}, got {None if src is None else tuple(src.shape)}"
    dst.copy_(src, non_blocking=True)
    if src.is_cuda:
        src.record_stream(stream)      # a caller that drops the tensor now must not see its memory reused under the copy


def _queues_warning(n, who):
    # the streams overlap only when each sits on a hardware queue (and pipe) of its own: the runtime multiplexes HIP streams
    # onto GPU_MAX_HW_QUEUES queues (default 4) in creation order and reads the variable once, at start-up
    q = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)
    if q < n:
        warnings.warn(f"{who}: {n} streams on GPU_MAX_HW_QUEUES={q} hardware queues -- streams will share queues and "
                      f"serialise; export GPU_MAX_HW_QUEUES>={n} before the process starts")


class StagedExtractor:
    """``pipe = StagedExtractor(ldm, example_inputs); outs, done = pipe.submit(batched_inputs)``.

    ``example_inputs`` fixes what the graphs are captured for: tensor shapes, the presence of ``cond_emb`` and the
    ``timestep`` range (a construction-time constant, like every other non-tensor entry).  ``submit`` takes a dict with the
    same keys as ``LdmRocm.forward`` (``img`` [B,3,H,W] f32, ``cond_inputs`` [B,77,768], ``cond_emb`` [B,1,1280] | None) on the
    pipeline's device, or in host memory (pinned for an asynchronous transfer: copied in on the encoder stream)."""

    def __init__(self, ldm, example_inputs, unet_streams=3, streams=None, sync_inputs=True, range_check=None, slots=None,
                 **kwargs):
        assert unet_streams >= 1
        self.ldm = ldm
        self.k = int(unet_streams)
        # slots >= UNet streams: slot j's UNet runs on stream j mod k; with more slots than streams the encoder stream may run
        # further ahead of the UNets (a slot's encoder waits for that slot's previous UNet only)
        self.n_slots = int(slots or unet_streams)
        assert self.n_slots >= self.k
        self.sync_inputs = bool(sync_inputs)
        _queues_warning(self.k + 1, "StagedExtractor")
        dev = example_inputs['img'].device
        self.device = dev
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
            # the encoder stream at high priority: its chip-filling kernels are the pipeline's pacemaker, and with six slots the
            # UNet streams always have work queued behind them (round 5, same box, two runs: 372.6 / 372.2 -> 377.6 / 377.7
            # images/s; UNet streams high instead: 370.2 / 369.6; round 3's three-slot pipeline: 307.9 vs 311.6 without).
            # MADM_EXP_PRIO: "e" (default) / "u" = UNet streams high / "0" = all equal
            prio = os.environ.get("MADM_EXP_PRIO", "e")
            self.s_enc = torch.cuda.Stream(device=dev, priority=-1 if prio == "e" else 0)
            self.s_unet = [torch.cuda.Stream(device=dev, priority=-1 if prio == "u" else 0) for _ in range(self.k)]
        # per-slot static inputs: the graphs read THESE tensors; everything that is not a tensor is a capture-time constant
        self.tensor_keys = [k_ for k_ in ('img', 'cond_inputs', 'cond_emb') if k_ in example_inputs]
        self.const_inputs = {k_: v for k_, v in example_inputs.items() if k_ not in self.tensor_keys}
        assert not any(torch.is_tensor(v) for v in self.const_inputs.values()), \
            f"StagedExtractor: unexpected tensor inputs {[k_ for k_, v in self.const_inputs.items() if torch.is_tensor(v)]}"
        self.static = []
        for j in range(self.n_slots):
            d = {k_: _static_like(example_inputs[k_]) for k_ in self.tensor_keys}
            assert d['img'].dtype == torch.float32, "img is handed over as f32 NCHW (the stem kernel normalises it)"
            d.update(self.const_inputs)
            self.static.append(d)
        self.enc_graphs, self.unet_graphs, self.slots, self.outs = [], [], [], []
        cur = torch.cuda.current_stream(dev)
        # the graphs hold the THROUGHPUT rows of the tile table (launches of several batches side by side): pinned, so that nothing
        # inside switches to the lone-launch profile of a synchronous forward (ops.tuning_profile)
        with torch.no_grad(), ops.tuning_profile("throughput", pin=True):
            for j in range(self.n_slots):
                self.s_enc.wait_stream(cur)
                with torch.cuda.stream(self.s_enc):
                    ldm._stage_encode(self.static[j])          # sizes this stream's workspaces outside the capture
                self.s_enc.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.s_enc):
                    st = ldm._stage_encode(self.static[j])
                self.enc_graphs.append(g)
                self.slots.append(st)
            for j in range(self.n_slots):
                s = self.s_unet[j % self.k]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    ops.ARENA.reset(dev)
                    ldm._stage_unet(self.slots[j], self.static[j], **kwargs)
                s.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    ops.ARENA.reset(dev)                       # the statistics arena of this stage: zeroed inside the graph
                    self.outs.append(ldm._stage_unet(self.slots[j], self.static[j], **kwargs))
                self.unet_graphs.append(g)
        torch.cuda.synchronize(dev)
        self.encoded = [torch.cuda.Event() for _ in range(self.n_slots)]
        self.done = [torch.cuda.Event() for _ in range(self.n_slots)]
        self._ready = [torch.cuda.Event() for _ in range(2 * self.n_slots)]
        self.turn = 0
        if range_check is None:
            range_check = bool(ldm.check_input_range)
        has_probe = self.slots[0].get("minmax") is not None
        self.range_check = DeferredRangeCheck(dev) if (range_check and has_probe) else None

    @property
    def streams(self):
        return [self.s_enc, *self.s_unet]

    def submit(self, batched_inputs):
        """Enqueues one batch; returns ``Submitted`` = (outputs, event): the slot's output tensors are valid once ``event`` has
        fired and stay so until the slot comes round again (``slots`` submits later; default = ``unet_streams``) -- consume them on the host after
        ``event.synchronize()`` or on a stream after ``stream.wait_event(event)``; a consumer that works on ANOTHER stream
        than the slot's UNet stream must have finished (or be waited for) before that later submit.

        The caller's tensors are read by copies enqueued on the ENCODER stream, behind an event recorded on the caller's
        current stream (``sync_inputs``: whatever produced them there is finished first).  Like the source of any asynchronous
        copy they must not be overwritten before the copies ran: ``.taken`` of the result fires then (dropping them is safe).
        Raises the deferred range ``AssertionError`` of an EARLIER batch when its probe has arrived."""
        for k_ in {*self.const_inputs, *batched_inputs} - set(self.tensor_keys):
            v, got = self.const_inputs.get(k_), batched_inputs.get(k_)
            assert not torch.is_tensor(got) and got == v, \
                f"StagedExtractor: '{k_}' = {v!r} is fixed at construction (captured in the graphs), got {got!r}"
        if self.range_check is not None:
            self.range_check.make_room()       # may raise an EARLIER batch's assert -- before anything of this submit is enqueued
        j = self.turn % self.n_slots
        first = self.turn < self.n_slots
        turn = self.turn
        self.turn += 1
        st = self.static[j]
        if self.sync_inputs:
            ready = self._ready[turn % len(self._ready)]
            ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.s_enc):
            if self.sync_inputs:
                self.s_enc.wait_event(ready)
            if not first:
                self.s_enc.wait_event(self.done[j])            # the slot's input and hand-over buffers are free again
            # host tensors (pinned: asynchronous) are transferred here too, in stream order in front of their encoder: measured
            # against a transfer stream of the pipeline's own with a staging ring, 2 x 3 x 512 x 512 f32 per step from pinned
            # memory, same box: 358.5 images/s from HBM, 323 .. 333 this way, 307 with the transfer stream (a fifth stream shares
            # a hardware pipe with one of the four working ones)
            for k_ in self.tensor_keys:
                _copy_checked(st[k_], batched_inputs.get(k_), k_, self.s_enc)
            taken = torch.cuda.Event()         # one per submit: the caller may hold it for as long as it likes
            taken.record(self.s_enc)
            self.enc_graphs[j].replay()
            self.encoded[j].record(self.s_enc)
            if self.range_check is not None:
                self.range_check.push(self.slots[j]["minmax"], self.s_enc, turn)
        s = self.s_unet[j % self.k]
        with torch.cuda.stream(s):
            s.wait_event(self.encoded[j])
            self.unet_graphs[j].replay()
            self.done[j].record(s)
        return Submitted((self.outs[j], self.done[j]), taken, turn)

    def drain(self):
        """Waits for everything submitted and runs the range checks still pending."""
        for s in self.streams:
            s.synchronize()
        if self.range_check is not None:
            self.range_check.drain()

    def quiesce(self):
        """Waits for the streams and DROPS the pending range checks (error paths: leave nothing in flight, raise nothing)."""
        for s in self.streams:
            s.synchronize()
        if self.range_check is not None:
            self.range_check.discard()

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


class GraphedInference:
    """``MadmInference.forward`` (mtmadise.py:657-691) as one hipGraph per slot; slots round-robin on ``streams`` streams, so
    ``streams`` images are in flight.  ``submit(batched_inputs)`` takes the reference's ``[{'target_second_modality': [3,H,W]}]``
    list (0 .. 255 values, any float / uint8 dtype, host or device), copies the image into the slot's static buffer and
    returns (``[{'sem_seg': [1,K,H,W]}]``, event).  One image size per runner (the graphs are captured for it); the
    extractor's range assert is deferred as in ``StagedExtractor``."""

    def __init__(self, model, example_inputs, streams=4, slots=None, sync_inputs=True, range_check=None):
        self.model = model
        self.n_streams = int(streams)
        self.n_slots = int(slots or streams)
        assert self.n_slots >= self.n_streams >= 1
        self.sync_inputs = bool(sync_inputs)
        _queues_warning(self.n_streams, "GraphedInference")
        assert len(example_inputs) == 1
        dev = next(model.parameters()).device
        self.device = dev
        ex = example_inputs[0]['target_second_modality']
        ldm = model.backbone.feature_extractor.ldm_extractor
        self.ldm = ldm
        self.streams_ = [torch.cuda.Stream(device=dev) for _ in range(self.n_streams)]
        self.static, self.graphs, self.outs, self.minmax = [], [], [], []
        cur = torch.cuda.current_stream(dev)
        with torch.no_grad(), ops.tuning_profile("throughput", pin=True):
            for j in range(self.n_slots):
                s = self.streams_[j % self.n_streams]
                x = torch.empty(tuple(ex.shape), dtype=torch.float32, device=dev).copy_(ex)
                call = [{'target_second_modality': x}]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    model(call)                     # sizes this stream's workspaces outside the capture
                s.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    out = model(call)
                self.static.append(x)
                self.graphs.append(g)
                self.outs.append(out)
                self.minmax.append(getattr(ldm, "last_minmax", None))
        torch.cuda.synchronize(dev)
        self.done = [torch.cuda.Event() for _ in range(self.n_slots)]
        self._ready = [torch.cuda.Event() for _ in range(2 * self.n_slots)]
        self.turn = 0
        if range_check is None:
            range_check = bool(ldm.check_input_range)
        self.range_check = DeferredRangeCheck(dev) if (range_check and self.minmax[0] is not None) else None

    def stream_of(self, slot):
        return self.streams_[slot % self.n_streams]

    def submit(self, batched_inputs):
        """Returns ``Submitted`` = (outputs, event, slot).  The outputs stay valid until the slot comes round again (``slots``
        submits later); work enqueued on ``stream_of(slot)`` before that is ordered in front of the slot's next replay by
        itself.  A device image must not be overwritten before ``.taken`` of the result has fired (see ``Submitted``)."""
        assert len(batched_inputs) == 1 and 'modality_type' not in batched_inputs[0]
        if self.range_check is not None:
            self.range_check.make_room()
        j = self.turn % self.n_slots
        turn = self.turn
        self.turn += 1
        s = self.stream_of(j)
        x = batched_inputs[0]['target_second_modality']
        assert tuple(x.shape) == tuple(self.static[j].shape), \
            f"GraphedInference: captured for images of shape {tuple(self.static[j].shape)}, got {tuple(x.shape)}"
        if self.sync_inputs and x.is_cuda:
            ready = self._ready[turn % len(self._ready)]
            ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            if self.sync_inputs and x.is_cuda:
                s.wait_event(ready)
            self.static[j].copy_(x, non_blocking=True)          # dtype conversion (uint8 -> f32) included
            if x.is_cuda:
                x.record_stream(s)
            taken = torch.cuda.Event()
            taken.record(s)
            self.graphs[j].replay()
            if self.range_check is not None:
                self.range_check.push(self.minmax[j], s, turn)
            self.done[j].record(s)
        return Submitted((self.outs[j], self.done[j], j), taken, turn)

    def drain(self):
        for s in self.streams_:
            s.synchronize()
        if self.range_check is not None:
            self.range_check.drain()

    def quiesce(self):
        """Waits for the streams and DROPS the pending range checks (error paths: leave nothing in flight, raise nothing)."""
        for s in self.streams_:
            s.synchronize()
        if self.range_check is not None:
            self.range_check.discard()


class StagedInference:
    """``MadmInference.forward`` (mtmadise.py:657-691) as THREE hipGraphs per slot -- VAE encoder | UNet | VAE decoder +
    projections + head -- the whole-model analogue of ``StagedExtractor``.  The two chip-filling stages run on
    streams of their own (encoder: high priority, the pipeline's pacemaker; decoder + head), the small-grid UNet stages of
    consecutive images side by side on ``unet_streams`` streams: four streams = the four hardware pipes of the queue scheduler.
    The stage boundaries are ``LdmRocm.stage_hook`` calls inside ONE ordinary ``model(batched_inputs)`` call: during capture the
    hook ends the running stream capture and begins the next one on the next stage's stream, so every module between the
    boundaries is captured exactly as the eager forward runs it (same kernels, same values: the slot outputs are bit-identical
    to ``model(batched_inputs)``).  ``submit`` has ``GraphedInference``'s contract."""

    def __init__(self, model, example_inputs, unet_streams=2, slots=None, sync_inputs=True, range_check=None):
        self.model = model
        self.k = int(unet_streams)
        self.n_slots = int(slots or 2 * (self.k + 1))
        assert self.k >= 1 and self.n_slots >= self.k
        self.sync_inputs = bool(sync_inputs)
        _queues_warning(self.k + 2, "StagedInference")
        assert len(example_inputs) == 1
        dev = next(model.parameters()).device
        self.device = dev
        ex = example_inputs[0]['target_second_modality']
        ldm = model.backbone.feature_extractor.ldm_extractor
        self.ldm = ldm
        self.s_enc = torch.cuda.Stream(device=dev, priority=-1)
        self.s_unet = [torch.cuda.Stream(device=dev) for _ in range(self.k)]
        self.s_dec = torch.cuda.Stream(device=dev)
        self.static, self.graphs, self.outs, self.minmax = [], [], [], []
        cur = torch.cuda.current_stream(dev)
        with torch.no_grad(), ops.tuning_profile("throughput", pin=True):
            for j in range(self.n_slots):
                stage_streams = [self.s_enc, self.s_unet[j % self.k], self.s_dec]
                x = torch.empty(tuple(ex.shape), dtype=torch.float32, device=dev).copy_(ex)
                call = [{'target_second_modality': x}]
                for s in stage_streams:
                    s.wait_stream(cur)
                self._run_staged(call, stage_streams, capture=False)      # sizes the three streams' workspaces outside the capture
                torch.cuda.synchronize(dev)
                out, graphs = self._run_staged(call, stage_streams, capture=True)
                self.static.append(x)
                self.graphs.append(graphs)
                self.outs.append(out)
                self.minmax.append(getattr(ldm, "last_minmax", None))
        torch.cuda.synchronize(dev)
        self.encoded = [torch.cuda.Event() for _ in range(self.n_slots)]
        self.unet_done = [torch.cuda.Event() for _ in range(self.n_slots)]
        self.done = [torch.cuda.Event() for _ in range(self.n_slots)]
        self._ready = [torch.cuda.Event() for _ in range(2 * self.n_slots)]
        self.turn = 0
        if range_check is None:
            range_check = bool(ldm.check_input_range)
        self.range_check = DeferredRangeCheck(dev) if (range_check and self.minmax[0] is not None) else None

    def _run_staged(self, call, streams, capture):
        """One ``model(call)`` whose three stages run on ``streams`` (eager: chained by events) or are captured into three graphs."""
        ldm, dev = self.ldm, self.device
        graphs, state = [], {"i": 0}
        prev = torch.cuda.current_stream(dev)

        def begin(i):
            torch.cuda.set_stream(streams[i])
            if capture:
                g = torch.cuda.CUDAGraph()
                g.capture_begin()
                graphs.append(g)

        def hook(_name):
            i = state["i"]
            if capture:
                graphs[i].capture_end()
            else:
                ev = torch.cuda.Event()
                ev.record(streams[i])
                streams[i + 1].wait_event(ev)
            state["i"] = i + 1
            begin(i + 1)

        assert ldm.__dict__.get("stage_hook") is None, "StagedInference: a stage hook is already installed on this extractor"
        ldm.__dict__["stage_hook"] = hook
        try:
            begin(0)
            out = self.model(call)
            assert state["i"] == 2, f"StagedInference: expected two stage boundaries in the forward, saw {state['i']}"
            if capture:
                graphs[2].capture_end()
        finally:
            ldm.__dict__["stage_hook"] = None
            torch.cuda.set_stream(prev)
        if not capture:
            for s in streams:
                s.synchronize()
        return out, graphs

    @property
    def streams_(self):
        return [self.s_enc, *self.s_unet, self.s_dec]

    def stream_of(self, slot):
        """The stream the slot's LAST stage runs on: work enqueued there after ``submit`` is ordered behind the outputs."""
        return self.s_dec

    def submit(self, batched_inputs):
        """Returns ``Submitted`` = (outputs, event, slot); see ``GraphedInference.submit``."""
        assert len(batched_inputs) == 1 and 'modality_type' not in batched_inputs[0]
        if self.range_check is not None:
            self.range_check.make_room()
        j = self.turn % self.n_slots
        first = self.turn < self.n_slots
        turn = self.turn
        self.turn += 1
        x = batched_inputs[0]['target_second_modality']
        assert tuple(x.shape) == tuple(self.static[j].shape), \
            f"StagedInference: captured for images of shape {tuple(self.static[j].shape)}, got {tuple(x.shape)}"
        if self.sync_inputs and x.is_cuda:
            ready = self._ready[turn % len(self._ready)]
            ready.record(torch.cuda.current_stream(self.device))
        g1, g2, g3 = self.graphs[j]
        with torch.cuda.stream(self.s_enc):
            if self.sync_inputs and x.is_cuda:
                self.s_enc.wait_event(ready)
            if not first:
                self.s_enc.wait_event(self.done[j])         # the slot's buffers (all three stages') are free again
            self.static[j].copy_(x, non_blocking=True)
            if x.is_cuda:
                x.record_stream(self.s_enc)
            taken = torch.cuda.Event()
            taken.record(self.s_enc)
            g1.replay()
            self.encoded[j].record(self.s_enc)
            if self.range_check is not None:
                self.range_check.push(self.minmax[j], self.s_enc, turn)
        su = self.s_unet[j % self.k]
        with torch.cuda.stream(su):
            su.wait_event(self.encoded[j])
            g2.replay()
            self.unet_done[j].record(su)
        with torch.cuda.stream(self.s_dec):
            self.s_dec.wait_event(self.unet_done[j])
            g3.replay()
            self.done[j].record(self.s_dec)
        return Submitted((self.outs[j], self.done[j], j), taken, turn)

    def drain(self):
        for s in self.streams_:
            s.synchronize()
        if self.range_check is not None:
            self.range_check.drain()

    def quiesce(self):
        for s in self.streams_:
            s.synchronize()
        if self.range_check is not None:
            self.range_check.discard()
