"""End-to-end parity of the HIP path (through the C ABI) with the committed golden vectors that the
reference's own functions produced (tests/golden/gen_golden.py), and with the live CPU oracle.

Tolerances (stated per north_star): exact-f32 mode: 1e-3 relative to the tensor's max magnitude
(observed ~1e-5); bf16 mode: relative L2 <= 4e-2 per tensor -- bf16 storage rounds every activation
to 8 bits of mantissa over ~60 layers, the reference itself runs fp16 autocast."""
import pytest
import torch

from golden_util import CASES, make_inputs, load_golden, add_lora, tap_subset
from util import rel_err

pytestmark = pytest.mark.gpu

F32_TOL = 1e-3
BF16_L2_TOL = 4e-2
F16_L2_TOL = 8e-3     # fp16 storage (10 mantissa bits): the reference's own autocast arithmetic (engine/train_loop.py:277)
# Per-tensor gates of the golden comparisons = 1.5x the worst value observed on MI355X over the four cases (round 6, VERDICT r5 #7:
# were 2x -- wide enough to hide the regression of one layer; profiles/round5_precision_f16_bf16.txt, unchanged since round 2;
# f32: max-relative 2.6e-6 .. 7.5e-6, gate ~15x that -- north_star's bound is 1e-3).  The arithmetic is deterministic: the same
# values on every box; a kernel change that moves a rounding pattern moves them by a few per cent, not by 50 %.
#            latents   sample    taps
GOLD_TOL = {torch.float32: dict(latents=1e-4, sample=1e-4, tap=1e-4),          # max |a - b| / max |b|
            torch.float16: dict(latents=3.0e-3, sample=3.3e-3, tap=3.75e-3),   # relative L2 (observed <= 2.02 / 2.21 / 2.49e-3)
            torch.bfloat16: dict(latents=2.25e-2, sample=2.9e-2, tap=2.9e-2)}  # relative L2 (observed <= 1.49 / 1.91 / 1.93e-2)


class _LoraConfig:
    def __init__(self, r, lora_alpha):
        self.r, self.lora_alpha = r, lora_alpha
        self.init_lora_weights = "gaussian"
        self.target_modules = ["to_k", "to_q", "to_v", "to_out.0"]


@pytest.fixture(scope="module")
def extractor(cuda):
    from madm_amd.ldm_rocm import LdmRocm
    m = LdmRocm("", encoder_block_indices=[], unet_block_indices=[5, 8, 11], decoder_block_indices=[],
                input_range='-1+1', unet_block_indices_type='after', finetune_unet='all',
                compute_dtype=torch.float32, weights='synthetic', seed=0)
    return m


def _run(m, case, dtype):
    images, cond_inputs, cond_emb, timesteps, _ = make_inputs(**case)
    m.compute_dtype = m.vae.compute_dtype = m.unet.compute_dtype = dtype
    t = case["t"]
    feats = m({"img": images.cuda(), "cond_inputs": cond_inputs.cuda(), "cond_emb": cond_emb.cuda(),
               "timestep": (t, t + 1)}, "rgb")
    torch.cuda.synchronize()
    sample = m.last_sample.nchw(4).cpu()
    return m.last_latents.cpu(), sample, [f.cpu() for f in feats]


def _compare(name, got, gold, dtype):
    lat, sample, feats = got
    report = []
    pairs = [("latents", lat, gold["latents"]), ("sample", sample, gold["sample"])]
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(gold[f"tap{i}_shape"].tolist()), (f.shape, gold[f"tap{i}_shape"])
        pairs.append((f"tap{i}", tap_subset(name, f), gold[f"tap{i}"]))
    ok = True
    for key, a, b in pairs:
        e, l2 = rel_err(a, b)
        report.append(f"{key}: max {e:.2e} l2 {l2:.2e}")
        tol = GOLD_TOL[dtype]["tap" if key.startswith("tap") else key]
        ok &= (e < tol) if dtype == torch.float32 else (l2 < tol)
    print(name, dtype, "; ".join(report))
    assert ok, f"{name} {dtype}: " + "; ".join(report)


@pytest.mark.parametrize("name", ["small_t0", "small_t60", "rect_t0", "full_t0"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_golden(extractor, name, dtype):
    _compare(name, _run(extractor, CASES[name], dtype), load_golden(name), dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_golden_lora(cuda, dtype):
    """peft-style LoRA (mtmadise.py:115-147): two adapters registered, 'Depth' active."""
    from madm_amd.ldm_rocm import LdmRocm
    m = LdmRocm("", [], [5, 8, 11], [], input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                compute_dtype=dtype, weights='synthetic', seed=0)
    add_lora(m.unet, _LoraConfig)
    names = [n for n, _ in m.unet.named_parameters() if "lora" in n]
    assert len(names) == 128 * 2 * 2 and all(("default" in n) or ("Depth" in n) for n in names)
    _compare("small_lora", _run(m, CASES["small_lora"], dtype), load_golden("small_lora"), dtype)
    # switching the adapter changes the result; disabling restores the base model
    base = load_golden("small_t60")
    for mod in m.unet.modules():
        if hasattr(mod, "_active_adapter"):
            mod._active_adapter = []
    _compare("small_t60", _run(m, CASES["small_t60"], dtype), base, dtype)


@pytest.mark.parametrize("name", ["small_t60", "full_t0"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_golden_vae_decoder_branch(extractor, name, dtype):
    """vae_decoder_loss branch of LdmDiffusers.forward (:192-201, all three shipped task configs): the
    decoded UNet sample replaces the encoder feature slot, and return_unet_final_output hands back
    {'before_vae.decoder', 'after_vae.decoder' (clipped)} (:211-215)."""
    m = extractor
    case = CASES[name]
    gold = load_golden(name)
    images, cond_inputs, cond_emb, timesteps, _ = make_inputs(**case)
    m.compute_dtype = m.vae.compute_dtype = m.unet.compute_dtype = dtype
    m.vae_decoder_loss = True
    try:
        t = case["t"]
        feats, extra = m({"img": images.cuda(), "cond_inputs": cond_inputs.cuda(), "cond_emb": cond_emb.cuda(),
                          "timestep": (t, t + 1)}, "rgb", return_unet_final_output=True)
        torch.cuda.synchronize()
    finally:
        m.vae_decoder_loss = False
    assert len(feats) == 4 and tuple(feats[0].shape) == tuple(gold["decoder_shape"].tolist())
    ds = 4 if name.startswith("full") else 1
    dec = feats[0].cpu()[:, :, ::ds, ::ds]
    e, l2 = rel_err(dec, gold["decoder"])
    e2, l22 = rel_err(extra["before_vae.decoder"].cpu(), gold["sample"])
    print(name, dtype, f"decoder max {e:.2e} l2 {l2:.2e}; sample max {e2:.2e}")
    if dtype == torch.float32:
        assert e < F32_TOL and e2 < F32_TOL
    else:
        assert l2 < 6e-2 and l22 < BF16_L2_TOL    # ~110 layers end to end in bf16
    clipped = extra["after_vae.decoder"].cpu()
    assert clipped.min() >= -1 and clipped.max() <= 1
    assert torch.equal(clipped, feats[0].cpu().clamp(-1, 1))
    for i, f in enumerate(feats[1:]):
        assert rel_err(tap_subset(name, f.cpu()), gold[f"tap{i}"])[1] < (1e-4 if dtype == torch.float32 else BF16_L2_TOL)


def test_batch_invariance_and_determinism(extractor):
    """Size-independent properties at the full 512x512 size in bf16: images are independent units
    (GroupNorm/LayerNorm are per-sample), so a batch of two equal images gives two equal outputs that
    equal the single-image result bit for bit, and repeated runs are bit-identical."""
    case = dict(CASES["full_t0"])
    images, cond_inputs, cond_emb, _, _ = make_inputs(**case)
    m = extractor
    m.compute_dtype = m.vae.compute_dtype = m.unet.compute_dtype = torch.bfloat16

    def run(B):
        feats = m({"img": images.repeat(B, 1, 1, 1).cuda(), "cond_inputs": cond_inputs.repeat(B, 1, 1).cuda(),
                   "cond_emb": cond_emb.repeat(B, 1, 1).cuda()}, "rgb")
        torch.cuda.synchronize()
        return [f.cpu() for f in feats]

    one, two, again = run(1), run(2), run(2)
    for a, b, c in zip(one, two, again):
        assert torch.equal(b[0:1], b[1:2]), "batch elements differ"
        assert torch.equal(b, c), "not deterministic"
        e, l2 = rel_err(b[0:1], a)
        # tile / split-K choices depend on M, so B=1 and B=2 are two different (equally valid) bf16
        # evaluations: they agree to bf16 noise, not bitwise
        assert l2 < 2.5e-2, f"B=2 vs B=1: {e:.2e} {l2:.2e}"



def _distinct_batches(n, B, H, W):
    """n DIFFERENT batches (image, prompt tokens and time-embedding residual all differ from batch to batch and from image to
    image): a slot that read another slot's input or hand-over buffer cannot reproduce forward() on its own batch."""
    out = []
    for i in range(n):
        g = torch.Generator().manual_seed(5000 + i)
        out.append({"img": torch.rand((B, 3, H, W), generator=g).cuda(),
                    "cond_inputs": (0.02 * torch.randn((B, 77, 768), generator=g)).cuda(),
                    "cond_emb": (0.02 * torch.randn((B, 1, 1280), generator=g)).cuda()})
    return out


def _forward_each(m, batches):
    from madm_amd import ops
    want = []
    # (the runners capture under the THROUGHPUT rows of the tile table; a plain forward() would take the lone-launch rows of the
    # latency profile -- other tiles / split-K, another summation order -- so the bit-for-bit reference runs pinned to the same rows)
    with torch.no_grad(), ops.tuning_profile("throughput", pin=True):
        for b in batches:
            want.append([f.clone() for f in m(b, "rgb")])
    torch.cuda.synchronize()
    for i in range(1, len(want)):        # the batches really differ: so do the results
        assert not torch.equal(want[i][0], want[0][0])
    return want


def _assert_slots_equal(got, want):
    bad = []
    for step, (feats, ref) in enumerate(zip(got, want)):
        assert len(feats) == len(ref)
        for i, (a, b) in enumerate(zip(feats, ref)):
            if not torch.equal(a, b):
                bad.append((step, i, int((a != b).sum()), float((a.float() - b.float()).abs().max())))
    assert not bad, f"staged pipeline differs from forward() on the same batch (step, tap, elements, max diff): {bad[:8]}"


@pytest.mark.parametrize("slots", [None, 5], ids=["slots3", "slots5"])
def test_staged_pipeline_matches_forward(extractor, slots):
    """madm_amd.pipeline.StagedExtractor (bench.py's launch strategy: encoder graphs on one stream, UNet graphs on K
    streams, K + 1 batches in flight) must hand back exactly what LdmRocm.forward returns for the batch that was SUBMITTED,
    for every slot and on every round: seven submits, seven different batches (``slots5``: more slots than UNet streams, two
    slots share a stream and its split-K workspace)."""
    from madm_amd.pipeline import StagedExtractor
    m = extractor
    m.compute_dtype = m.vae.compute_dtype = m.unet.compute_dtype = torch.float16
    batches = _distinct_batches(7, 2, 64, 64)
    want = _forward_each(m, batches)
    with torch.no_grad():
        pipe = StagedExtractor(m, batches[0], unet_streams=3, slots=slots)
        assert pipe.range_check is not None          # the reference's range assert stays on, deferred
        got = []
        for b in batches:                            # every slot is reused at least once
            outs, done = pipe.submit(b)
            done.synchronize()
            got.append([f.clone() for f in outs])
        pipe.drain()
        assert pipe.range_check.checked == len(batches)
        if slots is None:
            # the same batches from HOST memory (pinned image, pageable prompt tokens): copied in on the pipeline's transfer stream
            for b in batches:
                hb = {"img": b["img"].cpu().pin_memory(), "cond_inputs": b["cond_inputs"].cpu(), "cond_emb": b["cond_emb"]}
                outs, done = pipe.submit(hb)
                done.synchronize()
                got.append([f.clone() for f in outs])
            pipe.drain()
            want = want + want
    _assert_slots_equal(got, want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_staged_pipeline_matches_forward_at_512(extractor, dtype):
    """The TIMED launch path at the metric's size (BASELINE configs[1]: 2 x 3 x 512 x 512): four graphs of full-chip
    kernels running side by side -- the 16 x 16 halo conv of one batch's VAE encoder next to other batches' GEMM / attention
    / packed-f32 stem and LayerNorm kernels, two workgroups of different kernels per CU.  Twelve submits carry twelve
    DIFFERENT batches (images, prompts, time residuals); every slot of every round must be bit-identical to LdmRocm.forward on
    ITS batch (VERDICT r4 "What's weak" 2: with equal inputs in every slot, a slot reading another slot's hand-over buffer or an
    encoder overwriting a buffer a UNet still reads would go unseen; r3 item 2: a co-residency-triggered miscompute would
    show here too).  The submits are NOT separated by host syncs, so the batches really overlap; outputs are copied out on
    the slot's own stream behind its UNet.  The caller keeps ONE set of input tensors and refills it for every submit, on
    its own stream, behind the previous submit's ``taken`` event (the pipeline orders its copies behind the caller's fill by
    itself: ``sync_inputs``)."""
    from madm_amd.pipeline import StagedExtractor
    m = extractor
    m.compute_dtype = m.vae.compute_dtype = m.unet.compute_dtype = dtype
    batches = _distinct_batches(12, 2, 512, 512)
    want = _forward_each(m, batches)
    with torch.no_grad():
        pipe = StagedExtractor(m, batches[0], unet_streams=3, slots=6)       # bench.py's configuration
        assert pipe.range_check is not None
        got = []
        stage = {k: torch.empty_like(v) for k, v in batches[0].items()}
        cur = torch.cuda.current_stream()
        taken = None
        for step, b in enumerate(batches):          # every slot comes round twice, up to 7 batches in flight
            if taken is not None:
                cur.wait_event(taken)               # the source of an asynchronous copy: refill it once it has been read
            for k in stage:                         # the caller's buffers: ONE set, refilled for every submit
                stage[k].copy_(b[k])
            sub = pipe.submit(stage)
            (outs, done), taken = sub, sub.taken
            slot = step % pipe.n_slots
            s = pipe.s_unet[slot % pipe.k]
            with torch.cuda.stream(s):              # behind this slot's UNet, before the slot's next encoder may start
                got.append([f.clone() for f in outs])
                pipe.done[slot].record(s)
        pipe.drain()
        assert pipe.range_check.checked == len(batches)
    _assert_slots_equal(got, want)


def test_staged_pipeline_deferred_range_assert(extractor):
    """The reference asserts the input range on every call (ldm_diffusers.py:147).  The pipeline keeps the assert without the
    per-call host sync: an out-of-range batch raises AssertionError from a LATER submit or from drain(), naming the submit;
    in-range batches before and after it pass; shapes other than the captured ones and a changed timestep range are refused."""
    from madm_amd.pipeline import StagedExtractor
    m = extractor
    m.compute_dtype = m.vae.compute_dtype = m.unet.compute_dtype = torch.float16
    batches = _distinct_batches(5, 2, 64, 64)
    bad = dict(batches[2])
    bad["img"] = bad["img"].clone()
    bad["img"][1, 2, 7, 9] = 1.25                    # (1.25 - 0.5) / 0.5 = 1.5 > 1
    with torch.no_grad():
        pipe = StagedExtractor(m, batches[0], unet_streams=2)
        pipe.submit(batches[0])
        pipe.submit(batches[1])
        pipe.drain()
        pipe.submit(bad)                             # submit #2
        with pytest.raises(AssertionError, match=r"submit #2 .*max 1\.5"):
            for b in batches[3:]:
                pipe.submit(b)
            pipe.drain()
        pipe.drain()                                 # the pipeline stays usable: the later batches are checked and pass
        outs, done = pipe.submit(batches[4])
        pipe.drain()
        with pytest.raises(AssertionError, match="captured for shape"):
            pipe.submit({**batches[0], "img": batches[0]["img"][:1]})
        with pytest.raises(AssertionError, match="fixed at construction"):
            pipe.submit({**batches[0], "timestep": (60, 61)})
        # an unchecked pipeline can be asked for explicitly
        assert StagedExtractor(m, batches[0], unet_streams=1, range_check=False).range_check is None


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16], ids=["f32", "f16", "bf16"])
def test_fused_proj_out_matches_two_launch_path(extractor, dtype):
    """Transformer2DModel's fast path ([h2 | g] [Wp | Wp Wf]^T + (Wp bf + bp) + x, one launch) against the
    two-launch path it replaces (ff.net[2] + residual, then proj_out + residual), same weights, same input: equal up to the
    one extra rounding of h3 that the two-launch path has (f32: summation order only)."""
    from madm_amd import sd_unet
    m = extractor
    m.compute_dtype = m.vae.compute_dtype = m.unet.compute_dtype = dtype
    b = _distinct_batches(1, 2, 64, 64)[0]
    assert sd_unet.FUSE_PROJ_OUT
    with torch.no_grad():
        fused = [f.clone() for f in m(b, "rgb")]
        sd_unet.FUSE_PROJ_OUT = False
        try:
            plain = [f.clone() for f in m(b, "rgb")]
        finally:
            sd_unet.FUSE_PROJ_OUT = True
    tol = {torch.float32: 2e-5, torch.float16: 3e-3, torch.bfloat16: 2.5e-2}[dtype]
    for i, (a_, b_) in enumerate(zip(fused, plain)):
        e, l2 = rel_err(a_.cpu(), b_.cpu())
        print(dtype, f"tap{i}: fused vs two-launch max {e:.2e} l2 {l2:.2e}")
        assert l2 < tol, (i, e, l2)


def test_helper_functions_match_reference_signatures(extractor):
    """vae_encoder / add_noise / diffusion_unet keep the reference's call signatures
    (ldm_diffusers.py:283,349,454) and reproduce the golden intermediates when chained by hand."""
    from madm_amd import ldm_rocm
    m = extractor
    m.compute_dtype = m.vae.compute_dtype = m.unet.compute_dtype = torch.float32
    case = CASES["small_t60"]
    gold = load_golden("small_t60")
    images, cond_inputs, cond_emb, timesteps, shared_noise = make_inputs(**case)
    x = ((images - 0.5) / 0.5).cuda()
    latents, feats = ldm_rocm.vae_encoder(vae=m.vae, images=x, encoder_block_indices=[])
    assert feats == [] and rel_err(latents.cpu(), gold["latents"])[0] < F32_TOL
    noisy = ldm_rocm.add_noise(noise_scheduler=m.noise_scheduler, latents=latents, timesteps=timesteps.cuda(),
                               shared_noise=m.shared_noise)
    assert rel_err(noisy.cpu(), gold["noisy"])[0] < F32_TOL
    out, taps = ldm_rocm.diffusion_unet(unet=m.unet, sample=noisy, timestep=timesteps.cuda(),
                                        encoder_hidden_states=cond_inputs.cuda(), res_time_embedding=cond_emb.cuda(),
                                        unet_block_indices=[5, 8, 11], unet_block_indices_type='after')
    assert rel_err(out.sample.cpu(), gold["sample"])[0] < F32_TOL
    for i, t in enumerate(taps):
        assert rel_err(t.cpu(), gold[f"tap{i}"])[0] < F32_TOL
    # encoder taps are 1-based resnet counts (:291-293); 'in' taps are post-concat (:372-375)
    _, etaps = ldm_rocm.vae_encoder(vae=m.vae, images=x, encoder_block_indices=[5])
    assert tuple(etaps[0].shape) == (2, 512, 16, 16)
    _, itaps = ldm_rocm.diffusion_unet(unet=m.unet, sample=noisy, timestep=timesteps.cuda(),
                                       encoder_hidden_states=cond_inputs.cuda(), res_time_embedding=None,
                                       unet_block_indices=[0, 5, 11], unet_block_indices_type='in')
    assert [t.shape[1] for t in itaps] == [2560, 1920, 640]
    # vae_decoder: reference signature (:314), taps are taken BEFORE the indexed resnet (:330-333)
    dec, dtaps = ldm_rocm.vae_decoder(vae=m.vae, latents=torch.from_numpy(gold["sample"].numpy()).cuda(),
                                      decoder_block_indices=[0, 3], output_final=True)
    assert rel_err(dec.cpu(), gold["decoder"])[0] < F32_TOL
    assert [tuple(t.shape[1:]) for t in dtaps] == [(512, 8, 8), (512, 16, 16)]
    none, _ = ldm_rocm.vae_decoder(vae=m.vae, latents=latents, decoder_block_indices=[], output_final=False)
    assert none is None


_UNET_BWD_CACHE = {}


def _unet_backward_reference(lora):
    """CPU oracle + torch autograd, once per LoRA flag (about 20 s): inputs, loss gradients, reference gradients, and
    the HIP-side UNet holding the same synthetic weights."""
    if lora in _UNET_BWD_CACHE:
        return _UNET_BWD_CACHE[lora]
    from oracle import sd_modules, ldm_path
    from madm_amd import weights
    from madm_amd.sd_unet import UNet2DConditionModel
    B, hw, Lk, taps = 2, 8, 77, (5, 8, 11)
    ref_unet = sd_modules.UNet2DConditionModel()
    unet = UNet2DConditionModel()
    weights.synth_init_(ref_unet, 0, "unet.")
    weights.synth_init_(unet, 0, "unet.")
    if lora:
        add_lora(ref_unet, sd_modules.LoraConfig)
        add_lora(unet, _LoraConfig)
    g = torch.Generator().manual_seed(77)
    sample = torch.randn((B, 4, hw, hw), generator=g).requires_grad_(True)
    ctx = (0.5 * torch.randn((B, Lk, 768), generator=g)).requires_grad_(True)
    cond = (0.02 * torch.randn((B, 1280), generator=g)).requires_grad_(True)
    ts = torch.full((B,), 60, dtype=torch.int64)
    out, feats = ldm_path.diffusion_unet(ref_unet, sample, ts, ctx, cond, taps, "after")
    gs = [torch.randn(f.shape, generator=g) for f in feats]
    gout = torch.randn(out.sample.shape, generator=g)
    loss = sum((f * g_).sum() for f, g_ in zip(feats, gs)) + (out.sample * gout).sum()
    loss.backward()
    want = {n: p.grad for n, p in ref_unet.named_parameters() if p.grad is not None}
    _UNET_BWD_CACHE[lora] = dict(unet=unet.cuda(), sample=sample, ctx=ctx, cond=cond, ts=ts, gs=gs, gout=gout, want=want,
                                 B=B, hw=hw, Lk=Lk, taps=taps)
    return _UNET_BWD_CACHE[lora]


@pytest.mark.parametrize("lora", [True, False], ids=["lora", "plain"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_unet_backward_equals_oracle_autograd(cuda, dtype, lora):
    """backward.unet_backward (the whole SD-v1-4 UNet at full width on an 8x8 latent: every ResnetBlock2D /
    Transformer2DModel / down- and up-sampler / time-embedding / cross-attention K/V gradient composed from the C-ABI
    kernels, interiors recomputed) against torch autograd through the CPU oracle's diffusion_unet for a loss that is
    linear in the three tapped features and the final sample: gradients of the latents, the prompt tokens, the
    time-embedding residual and EVERY parameter (peft-style LoRA active on q / k / v / out in the 'lora' case).
    Tolerances as for the forward goldens: f32 mode 1e-3 of the tensor's max magnitude; bf16 mode relative L2 against
    the SAME f32 reference (bf16 storage of weights and activations through ~120 layers forward + backward)."""
    from madm_amd import backward, ops
    from madm_amd.nn import Tok
    from util import to_tokens, from_tokens
    r = _unet_backward_reference(lora)
    B, hw, Lk = r["B"], r["hw"], r["Lk"]
    kt = ops.k_tile(dtype)
    x = Tok(to_tokens(r["sample"].detach(), dtype, kt), B, hw, hw)
    ctx_d = r["ctx"].detach().reshape(B * Lk, 768).to(dtype).cuda()
    res = backward.unet_backward(r["unet"], x, r["ts"].cuda(), ctx_d, Lk, [to_tokens(g_, dtype) for g_ in r["gs"]],
                                 r["taps"], cond_emb=r["cond"].detach().cuda(), dsample=to_tokens(r["gout"], dtype))
    torch.cuda.synchronize()
    want = r["want"]
    BF16_GRAD_L2 = 8e-2   # observed worst over the 945 tensors: 4.5e-2 (f32 mode: 6.5e-6 max-relative)

    def check(name, got, ref_t):
        # the mid block runs on a 1x1 map here: self-attention over ONE key has softmax == 1, so dq = dk = 0 and the
        # to_q / to_k gradients of that layer are rounding noise on both sides -- measured against to_v's scale
        if "mid_block" in name and ".attn1.to_" in name and (".to_q." in name or ".to_k." in name):
            scale = float(want[name.replace(".to_q.", ".to_v.").replace(".to_k.", ".to_v.")].abs().max())
            worst = max(float(got.abs().max()), float(ref_t.abs().max())) / scale
            return worst < (1e-3 if dtype == torch.float32 else 5e-2), f"{name}: zero-gradient layer, {worst:.2e} of to_v l2 0"
        e, l2 = rel_err(got, ref_t)
        ok = (e < F32_TOL) if dtype == torch.float32 else (l2 < BF16_GRAD_L2)
        return ok, f"{name}: max {e:.2e} l2 {l2:.2e}"

    results = [check("d sample", from_tokens(res["sample"], B, hw, hw)[:, :4], r["sample"].grad),
               check("d ctx", res["ctx"].float().cpu()[:, :768].reshape(B, Lk, 768), r["ctx"].grad),
               check("d cond_emb", res["cond_emb"].cpu(), r["cond"].grad)]
    assert set(res["grads"]) == set(want), sorted(set(res["grads"]) ^ set(want))[:10]
    for n, gr in want.items():
        results.append(check(n, res["grads"][n].float().cpu().reshape(gr.shape), gr))
    bad = [msg for ok, msg in results if not ok]
    ranked = sorted(results, key=lambda t: -float(t[1].split("l2 ")[1]))
    print(dtype, "lora" if lora else "plain", len(results), "tensors; worst three:", [m for _, m in ranked[:3]])
    assert not bad, bad[:8]


def test_ldm_rocm_is_differentiable_through_autograd(cuda):
    """the extractor is an nn.Module whose output is differentiable w.r.t. its requires_grad parameters
    and the conditioning inputs through standard autograd.  LdmRocm.forward under grad mode returns taps that hang on
    ONE autograd node (_UNetTapsFn) whose backward is backward.unet_backward: loss.backward() fills .grad of the active
    LoRA matrices, the prompt tokens and the time-embedding residual with exactly what the explicit call returns."""
    from madm_amd import backward, ops
    from madm_amd.ldm_rocm import LdmRocm
    from madm_amd.nn import Tok
    m = LdmRocm("", [], [5, 8, 11], [], input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                compute_dtype=torch.float32, weights='synthetic', seed=0)
    add_lora(m.unet, _LoraConfig)
    for n, p in m.unet.named_parameters():
        p.requires_grad = ".lora_" in n and ".Depth." in n
    case = CASES["small_lora"]
    images, cond_inputs, cond_emb, _, _ = make_inputs(**case)
    ci = cond_inputs.cuda().requires_grad_(True)
    ce = cond_emb.cuda().requires_grad_(True)
    t = case["t"]
    inputs = {"img": images.cuda(), "cond_inputs": ci, "cond_emb": ce, "timestep": (t, t + 1)}
    feats = m(inputs, "rgb")
    assert len(feats) == 3 and all(f.requires_grad for f in feats)
    _compare("small_lora", (m.last_latents.cpu(), m.last_sample.nchw(4).cpu(), [f.detach().cpu() for f in feats]),
             load_golden("small_lora"), torch.float32)     # same values as the no-grad forward
    g = torch.Generator().manual_seed(5)
    gs = [torch.randn(f.shape, generator=g).cuda() for f in feats]
    sum((f * g_).sum() for f, g_ in zip(feats, gs)).backward()

    with torch.no_grad():
        st = m._stage_encode(inputs)
        B, h, w = st["B"], st["h"], st["w"]
        Lk = cond_inputs.shape[1]
        ctx = ops.cast_from_f32(cond_inputs.cuda().view(B * Lk, 768), torch.float32)
        res = backward.unet_backward(m.unet, Tok(st["noisy"], B, h, w), st["timesteps"], ctx, Lk,
                                     [ops.nchw_to_nhwc(g_, torch.float32, g_.shape[1]) for g_ in gs], (5, 8, 11),
                                     cond_emb=cond_emb.cuda()[:, 0].contiguous(), base_grads=False)
    torch.cuda.synchronize()
    assert rel_err(ci.grad.cpu(), res["ctx"].cpu()[:, :768].reshape(ci.shape))[0] < 1e-5
    assert rel_err(ce.grad.cpu(), res["cond_emb"].cpu().reshape(ce.shape))[0] < 1e-5
    n_checked = 0
    for n, p in m.unet.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, n
            if "mid_block" in n and ".attn1.to_" in n and (".to_q." in n or ".to_k." in n):
                # 1x1 map: softmax over one key, dq = dk = 0 -- rounding noise whose low bits depend on atomics order
                assert float(p.grad.abs().max()) < 1e-4 and float(res["grads"][n].abs().max()) < 1e-4, n
            else:
                assert rel_err(p.grad.cpu(), res["grads"][n].cpu().reshape(p.shape))[0] < 1e-5, n
            n_checked += 1
        else:
            assert p.grad is None, n
    assert n_checked == 128 * 2
    # without grad mode nothing is recorded
    with torch.no_grad():
        feats2 = m(inputs, "rgb")
    assert not any(f.requires_grad for f in feats2)


def test_extractor_training_step_lora(cuda):
    """train.ExtractorTrainer: HIP forward -> torch loss -> autograd through the UNet node -> flat-buffer clip + AdamW +
    EMA (engine/train_loop.py:203-217, cmdise.py:337-349).  Only the active LoRA matrices move, the first update equals
    torch.optim.AdamW on the same gradients, and a few steps on a fixed batch reduce the loss."""
    from madm_amd.ldm_rocm import LdmRocm
    from madm_amd.train import ExtractorTrainer
    m = LdmRocm("", [], [5, 8, 11], [], input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                compute_dtype=torch.float32, weights='synthetic', seed=0)
    add_lora(m.unet, _LoraConfig)
    for n, p in m.unet.named_parameters():
        p.requires_grad = ".lora_" in n and ".Depth." in n
    frozen = {n: p.detach().clone() for n, p in m.unet.named_parameters() if not p.requires_grad and "down_blocks.0" in n}
    case = CASES["small_lora"]
    images, cond_inputs, cond_emb, _, _ = make_inputs(**case)
    t = case["t"]
    batch = {"img": images.cuda(), "cond_inputs": cond_inputs.cuda(), "cond_emb": cond_emb.cuda(), "timestep": (t, t + 1)}
    with torch.no_grad():
        target = [f.clone() * 0.9 for f in m(batch, "rgb")]

    def loss_fn(feats):
        return sum(((f - g_) ** 2).mean() for f, g_ in zip(feats, target))

    lr = 2e-5   # Adam moves every element by ~lr per step: small against the LoRA entries (~0.04), linear regime
    tr = ExtractorTrainer(m, lr=lr, weight_decay=0.0, clip_grad=1.0, ema_alpha=0.9)
    before = tr.flat.flat.clone()
    l0, norm = tr.step(batch, loss_fn)
    assert norm is not None and norm > 0 and float(tr.flat.grad.abs().max()) > 0
    # first AdamW step with clipping == torch.optim.AdamW on the same (clipped) gradient
    ref_p = before.clone().requires_grad_(True)
    ref_p.grad = tr.flat.grad.clone() * min(1.0, 1.0 / (norm + 1e-6))
    torch.optim.AdamW([ref_p], lr=lr, weight_decay=0.0).step()
    assert rel_err((tr.flat.flat - before).cpu(), (ref_p.detach() - before).cpu())[0] < 1e-3   # the update itself
    # CMDISE._update_ema: alpha_teacher = min(1 - 1 / (iter + 1), ema_alpha) = 0.5 at the first update (cmdise.py:337-338)
    assert rel_err(tr.ema.cpu(), (0.5 * before + 0.5 * tr.flat.flat).cpu())[0] < 1e-6
    losses = [l0] + [tr.step(batch, loss_fn)[0] for _ in range(4)]
    print("losses", [f"{v:.5f}" for v in losses])
    assert losses[1] < losses[0] and losses[-1] < losses[0], losses
    for n, p in m.unet.named_parameters():
        if n in frozen:
            assert torch.equal(p.detach(), frozen[n]), n


def test_bench_workloads_have_tuned_rows(cuda):
    """VERDICT r5 #7 / DESIGN 12.3: the eval forward ran 2 x too long on two B = 1 shapes for a whole round because the launches
    had silently left the tuned table.  Every forward conv / linear launch of the bench workloads at their shipped sizes --
    extract (configs[1], f16 and bf16 share the rows), extract + one r = 8 adapter, eval (configs[2]), sliding windows
    (configs[4] geometry) -- must be decided by a ROW of madm_amd/csrc/igemm_tuned.inc (madm_conv2d_has_tuned_row), not by the
    fall-through heuristics; the message lists the missing rows in the table's own format (pinned to today's choice) so
    that a kernel change that moves shapes is followed by a re-tune (tools/tune_concurrent.py) or an explicit row."""
    import bench
    from types import SimpleNamespace
    from madm_amd import ops, weights
    from madm_amd.ldm_rocm import LdmRocm
    dev = torch.device("cuda")
    misses = {}

    def collect(tag, fn):
        ops.TILE_LOG = []
        try:
            with torch.no_grad():
                fn()
            torch.cuda.synchronize()
            for desc, tile, sk, has_row, flop in ops.TILE_LOG:
                if not has_row:
                    misses.setdefault(desc, [tile, sk, flop, set()])[3].add(tag)
            n = len(ops.TILE_LOG)
        finally:
            ops.TILE_LOG = None
        assert n > 50, (tag, n)

    dt = torch.float16
    m = LdmRocm("", encoder_block_indices=[], unet_block_indices=[5, 8, 11], decoder_block_indices=[], input_range='-1+1',
                unet_block_indices_type='after', finetune_unet='no', compute_dtype=dt, weights='synthetic', seed=0, device=dev)
    batch = bench.make_input_pool(2, 512, dev)[0]
    collect("extract", lambda: m(batch, "rgb"))
    m.unet.add_adapter(SimpleNamespace(r=8, lora_alpha=8, target_modules=["to_k", "to_q", "to_v", "to_out.0"]), "Depth")
    m.unet.set_adapter(["Depth"])
    weights.randomize_lora_B_(m.unet)
    collect("extract+lora", lambda: m(batch, "rgb"))
    del m
    torch.cuda.empty_cache()
    for tag, slide in (("eval", False), ("slide", True)):
        model = bench.build_eval_model(dt, dev, slide=slide, num_classes=9 if slide else 11)
        img = 255.0 * torch.rand((3, 512, 1024 if slide else 512), generator=torch.Generator().manual_seed(3))
        collect(tag, lambda: model([{"target_second_modality": img.to(dev)}]))
        del model
        torch.cuda.empty_cache()
    if misses:
        import re
        rows = []
        for desc, (tile, sk, flop, tags) in sorted(misses.items(), key=lambda kv: -kv[1][2]):
            g = re.match(r"dt(\d) M(\d+) N(\d+) K(\d+) k(\d) s(\d)( up)?( gn)?", desc)
            var = 1 if g.group(8) else (2 if g.group(7) else 0)
            rows.append(f"{{1, {g.group(2)}, {g.group(3)}, {g.group(4)}, {g.group(5)}, {var}, {tile}, {sk}}},   "
                        f"// {'/'.join(sorted(tags))}: {flop / 1e9:.2f} GFLOP, no row (heuristic choice pinned)")
        raise AssertionError(f"{len(rows)} launch shapes of the bench workloads are not in the tuned table:\n" + "\n".join(rows))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32], ids=["f16", "f32"])
def test_latency_and_throughput_profiles_agree_at_bench_size(extractor, dtype):
    """Round 6: a synchronous forward() consults the lone-launch rows of the tile / split-K table (latency profile), the
    runners the throughput rows.  Other tiles, other split-K factors -- the same sums in another order: at the bench's own size
    (2 x 3 x 512 x 512, where the 58 latency rows apply) the two profiles must agree to summation-order noise, in the timed
    arithmetic (f16), and the profiles really differ there (120 launches take another tile / split-K); the exact-f32 mode has no rows of its
    own in either table and must be bit-identical under both."""
    from madm_amd import ops
    m = extractor
    m.compute_dtype = m.vae.compute_dtype = m.unet.compute_dtype = dtype
    b = _distinct_batches(1, 2, 512, 512)[0]
    res, tiles = {}, {}
    for name in ("throughput", "latency"):
        ops.TILE_LOG = []
        try:
            with torch.no_grad(), ops.tuning_profile(name, pin=True):
                res[name] = [f.float().clone() for f in m(b, "rgb")]
            torch.cuda.synchronize()
            tiles[name] = [(d, t, sk) for d, t, sk, _, _ in ops.TILE_LOG]
        finally:
            ops.TILE_LOG = None
    assert len(tiles["latency"]) == len(tiles["throughput"]) > 100
    moved = sum(1 for a_, b_ in zip(tiles["latency"], tiles["throughput"]) if a_ != b_)
    if dtype == torch.float32:      # the tables hold rows of the 16-bit modes only: the f32 mode takes the same launches under both
        assert moved == 0 and all(torch.equal(x, y) for x, y in zip(res["latency"], res["throughput"]))
        return
    assert moved >= 10, f"only {moved} launches differ between the two profiles"
    # f16 storage: every layer output is rounded once, and a different summation order decorrelates those roundings -- two f16 forwards
    # of ~60 layers sit ~sqrt(2) x (the per-forward storage noise of ~1e-3) apart (observed 1.5e-3, MI355X; the golden gate of one
    # forward against fp32 is 3.75e-3)
    tol = 3e-3
    for i, (x, y) in enumerate(zip(res["latency"], res["throughput"])):
        e, l2 = rel_err(x.cpu(), y.cpu())
        print(dtype, f"tap{i}: latency vs throughput profile max {e:.2e} l2 {l2:.2e} ({moved} launches take another tile / split-K)")
        assert l2 < tol, (i, e, l2)
