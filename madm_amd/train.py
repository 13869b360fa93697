"""Training loop pieces around ``MTMADISE`` (BASELINE config 4): the reference's ``AMPTrainer.run_step``
(/root/reference/engine/train_loop.py:257-311) -- autocast forward, ``model.zero_grad()``, ``scaler.scale(losses)
.backward()``, unscale, ``clip_grad_norm_``, ``optimizer.step()``, ``scaler.update()`` -- on flat storage:

    losses = model(data)                      HIP forward (3 passes), loss scalars = outputs of ONE autograd node
    (scale * sum(losses)).backward()          explicit HIP backward; gradients accumulate into ONE flat fp32 buffer
    all-reduce(mean)                          dist.GradBucketReducer over that buffer (DDP's only job; world > 1)
    clip + AdamW                              optim.TableAdamW: ONE launch, per-tensor lr / weight decay / step
                                              (get_default_optimizer_params_unet: no decay on norms and biases, unet_lr)
    GradScaler bookkeeping                    dynamic loss scale (fp16 compute mode), inf / nan steps skipped

``ExtractorTrainer`` (round 1) trains the extractor alone against a caller-supplied torch loss and is kept for that use.
"""
import torch

from . import optim
from .dist import GradBucketReducer


class MadmTrainer:
    """One process per GPU.  ``dist``: an initialised torch.distributed module (backend nccl == RCCL, or gloo) or None.
    At construction rank 0's parameters are broadcast (DistributedDataParallel's start-up contract, main.py:289-294)."""

    def __init__(self, model, lr, weight_decay, grad_clip=None, unet_lr=None, betas=(0.9, 0.999), eps=1e-8, dist=None,
                 amp=True, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000,
                 lr_multiplier=None, exchange="allreduce", wire_dtype=None):
        self.model = model
        try:      # the extractor whose per-pass range assert this trainer defers to the end of its step
            self._ldm = model.backbone.feature_extractor.ldm_extractor
        except AttributeError:
            self._ldm = None
        table = optim.default_optimizer_params(model, lr, weight_decay, weight_decay_norm=0.0, weight_decay_bias=0.0,
                                               unet_lr=unet_lr)
        assert table, "nothing to train"
        # flat-buffer order = the order in which the explicit backward FINISHES gradients, reversed: parameters whose
        # gradient only exists at the very end of the backward (prompt / time gates, the time-embedding MLP and the
        # stacked time_emb_proj / batched cross-attention K/V projections, backward.unet_backward_from_state's tails) go
        # first, everything else follows in registration order (conv_in ... up blocks, projections, head) -- the backward
        # walks that part back to front, so finished gradients always form a growing TAIL of the buffer
        names = {id(p): n for n, p in model.named_parameters()}

        def late(p):
            n = names[id(p)]
            return ("clip_project" in n or ".time_embedding." in n or ".time_emb_proj." in n or
                    (".attn2.to_k." in n or ".attn2.to_v." in n))

        table = [t for t in table if late(t[0])] + [t for t in table if not late(t[0])]
        self.opt = optim.TableAdamW(table, betas=betas, eps=eps)
        self.grad_clip = grad_clip
        self.dist = dist
        self.reducer = GradBucketReducer(self.opt.flat.grad, dist, mode=exchange, wire_dtype=wire_dtype)
        self._index = {id(p): i for i, p in enumerate(self.opt.flat.params)}
        self._final = [False] * len(self.opt.flat.params)
        self._ptr = len(self._final) - 1
        self._pend_dst, self._pend_src, self._pend_ids = [], [], set()
        self.overlap = self.reducer.active     # world > 1 (or a forced single rank: dist.GradBucketReducer.active)
        model.grad_sink = self
        self.reduced_during_backward = 0       # elements whose all-reduce started before the backward returned
        if self.reducer.active:
            dist.broadcast(self.opt.flat.flat, src=0)
            torch.autograd.graph.increment_version(self.opt.flat.params)
            # DDP's _sync_module_states covers EVERY parameter and buffer, frozen ones included: the EMA teacher
            # (ema_sem_seg_head, ema_feature_projections, ema_clip_project_others) is requires_grad=False and, with
            # detectron2's seed + rank, starts from a different initialisation on every rank otherwise
            mine = {id(p) for p in self.opt.flat.params}
            frozen = [p for p in model.parameters() if id(p) not in mine]
            for p in frozen:
                dist.broadcast(p.data, src=0)
            if frozen:
                torch.autograd.graph.increment_version(frozen)     # packed-operand caches key on the version counter
            for b in model.buffers():               # (BatchNorm running statistics)
                dist.broadcast(b, src=0)
        # torch.cuda.amp.GradScaler semantics (only needed for the fp16 compute mode; harmless otherwise)
        self.scale = float(init_scale) if amp else 1.0
        self.amp, self.growth_factor, self.backoff_factor, self.growth_interval = amp, growth_factor, backoff_factor, growth_interval
        self._growth_tracker = 0
        self.lr_multiplier = lr_multiplier          # callable(iter) -> factor (WarmupParamScheduler in the shipped config)
        self.iter = 0
        self.last_allreduce_exposed_ms = None
        self.last_overlap_frac = None

    # ---- gradient sink of MTMADISE's autograd node: gradients land in the flat buffer, finished tails are reduced ----
    def final(self, p, g):
        i = self._index[id(p)]
        assert not self._final[i], "a gradient arrived after its span was handed to the all-reduce"
        # adds are batched (torch._foreach_add_: one launch per group of same-layout tensors instead of one per tensor --
        # 1 500 five-microsecond kernels per step); flushed before a span is handed to the all-reduce and at the end
        if g is not None:
            self._queue(p, g)
        self._final[i] = True
        if not self.overlap:
            if len(self._pend_dst) >= 256:
                self.flush()
            return
        ptr = self._ptr
        while ptr >= 0 and self._final[ptr]:
            ptr -= 1
        if ptr != self._ptr:
            lo = self.opt.flat.offsets[ptr + 1]
            if self.reducer.done_lo - lo >= self.reducer.bucket or ptr < 0:   # whole buckets only: few, large messages
                self.flush()
                self.reduced_during_backward += self.reducer.done_lo - lo
                self.reducer.reduce_tail(lo)
            self._ptr = ptr

    def accumulate(self, p, g):
        """A contribution that is not the parameter's last one this step (the source pass of the training step)."""
        assert not self._final[self._index[id(p)]]
        self._queue(p, g)
        if not self.overlap and len(self._pend_dst) >= 256:
            self.flush()

    def _queue(self, p, g):
        if id(p) in self._pend_ids:            # one destination twice in a multi-tensor add would race
            self.flush()
        g = g.reshape(p.shape) if g.dtype == p.grad.dtype else g.reshape(p.shape).to(p.grad.dtype)
        if not g.is_contiguous():              # permuted view of a 3 x 3 weight gradient: one strided add; a single
            p.grad.add_(g)                     # such tensor would push the whole multi-tensor call onto its slow path
            return
        self._pend_ids.add(id(p))
        self._pend_dst.append(p.grad)
        self._pend_src.append(g)

    def flush(self):
        if self._pend_dst:
            torch._foreach_add_(self._pend_dst, self._pend_src)
            self._pend_dst, self._pend_src = [], []
            self._pend_ids = set()

    def backward_done(self):
        self.flush()

    def run_step(self, data):
        """Returns (loss dict of python floats, total gradient norm, stepped)."""
        model = self.model
        assert model.training, "[MadmTrainer] model was changed to eval mode!"
        self.opt.zero_grad()
        self._final = [False] * len(self._final)
        self._ptr = len(self._final) - 1
        self.reduced_during_backward = 0
        ldm = self._ldm
        if ldm is not None and ldm.check_input_range:
            ldm.deferred_range_probes = []        # the three passes' range asserts (ldm_diffusers.py:147): checked at the step's end
        try:
            loss_dict = model(data)
        finally:
            probes = [] if ldm is None else (ldm.deferred_range_probes or [])
            if ldm is not None:
                ldm.deferred_range_probes = None
        losses = sum(loss_dict.values())
        (losses * self.scale).backward()
        self.flush()
        ev = None
        if self.overlap and torch.cuda.is_available() and self.opt.flat.grad.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self.reducer.finish()
        if ev is not None:
            ev[1].record()
        if self.lr_multiplier is not None:
            self.opt.lr_factor = float(self.lr_multiplier(self.iter))
        touched = getattr(model, "last_grad_param_ids", None)

        def check_probes():
            # the reference asserts before the batch is used (ldm_diffusers.py:147); here the three passes' probes are read at
            # the step's first host sync -- forward and backward have run, but NOTHING persistent has consumed the batch yet:
            # the optimizer step, the loss-scale update and the next step's EMA update are all behind this point
            for i, mm in enumerate(probes):
                lo, hi = mm.tolist()
                assert -1 <= lo and hi <= 1, \
                    f"input range check (ldm_diffusers.py:147), forward pass {i} of this step: min {lo} max {hi}"

        norm, stepped = self.opt.step(clip_grad=self.grad_clip, loss_scale=self.scale, touched=touched, on_synced=check_probes)
        if self.amp:
            if not stepped:
                self.scale *= self.backoff_factor
                self._growth_tracker = 0
            else:
                self._growth_tracker += 1
                if self._growth_tracker == self.growth_interval:
                    self.scale *= self.growth_factor
                    self._growth_tracker = 0
        self.iter += 1
        if ev is not None:      # all-reduce time the compute stream had to WAIT for after the backward (the exposed part)
            ev[1].synchronize()
            self.last_allreduce_exposed_ms = ev[0].elapsed_time(ev[1])
            self.last_overlap_frac = self.reduced_during_backward / max(1, self.opt.flat.numel)
        out = {k: float(v.detach()) for k, v in loss_dict.items()}       # (the step's host sync)
        return out, norm, stepped


class ExtractorTrainer:
    """One optimisation step of the LdmRocm extractor alone with a torch-side loss on its features (round-1 slice)."""

    def __init__(self, ldm, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, clip_grad=None, dist=None,
                 ema_alpha=None, extra_params=()):
        self.ldm = ldm
        params = [p for p in ldm.unet.parameters() if p.requires_grad] + [p for p in extra_params if p.requires_grad]
        assert params, "nothing to train: set requires_grad on the UNet / LoRA parameters first"
        self.flat = optim.FlatParams(params, with_grad=True)
        self.opt = optim.FlatAdamW(self.flat, lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.clip_grad = clip_grad
        self.reducer = GradBucketReducer(self.flat.grad, dist)
        self.ema_alpha = ema_alpha
        self.ema = self.flat.flat.clone() if ema_alpha is not None else None
        self.iter = 0

    def step(self, batched_inputs, loss_fn, input_modal="rgb", **kwargs):
        """Returns (loss value as a python float, total gradient norm or None)."""
        self.flat.grad.zero_()
        feats = self.ldm(batched_inputs, input_modal, **kwargs)
        loss = loss_fn(feats)
        loss.backward()
        self.reducer.finish()
        norm = self.opt.step(clip_grad=self.clip_grad)
        if self.ema is not None:   # CMDISE._update_ema: alpha_teacher = min(1 - 1 / (iter + 1), ema_alpha) (cmdise.py:337-338)
            self.iter += 1
            optim.ema_update(self.ema, self.flat.flat, min(1 - 1 / (self.iter + 1), self.ema_alpha))
        return float(loss.detach()), norm
