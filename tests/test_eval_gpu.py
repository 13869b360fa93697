"""Backbone projections, DAFormer head and the full inference forward (BASELINE config 3) on the GPU through the
C ABI, against (a) torch-CPU ops / the oracle restatements at small sizes and (b) the committed golden vectors that
the REFERENCE's own BasePromptTimeGenerator / AttentionFeatureExtractorBackbone / DAFormerHead classes produced
(tests/golden/gen_golden.py::main_eval).  Index ops (argmax labels) are compared bit-exact."""
import math

import pytest
import torch
import torch.nn.functional as F

from golden_util import EVAL_CASES, eval_image, init_eval_params, load_golden, seed_lora_
from util import to_tokens, from_tokens, rel_err, bf16_round

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


def _q(x, dtype):
    return bf16_round(x) if dtype == torch.bfloat16 else x


def _gen(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_spatial_kernels(cuda, dtype):
    from madm_amd import ops
    kt = ops.k_tile(dtype)
    # bilinear resize on tokens (up and down, into a column window of a wider buffer)
    x = _q(_gen((2, 64, 5, 7), 1), dtype)
    for (oh, ow) in ((20, 28), (3, 4), (5, 7), (16, 9)):
        ref = F.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=False)
        buf = torch.zeros((2 * oh * ow, 192), dtype=dtype, device="cuda")
        ops.resize_bilinear(to_tokens(x, dtype), 2, 5, 7, oh, ow, out=buf[:, 64:128])
        got = from_tokens(buf[:, 64:128], 2, oh, ow)
        assert rel_err(got, ref)[0] < (1e-6 if dtype == torch.float32 else 8e-3)
        assert buf[:, :64].abs().max().item() == 0 and buf[:, 128:].abs().max().item() == 0
    img = torch.rand((2, 3, 30, 50), generator=torch.Generator().manual_seed(2))
    assert rel_err(ops.resize_bilinear_nchw(img.cuda(), 64, 64).cpu(),
                   F.interpolate(img, size=(64, 64), mode="bilinear", align_corners=False))[0] < 1e-6
    # scale + zero pad / crop
    p = ops.scale_pad_nchw(img.cuda(), 1 / 255.0, 64, 64).cpu()
    assert torch.equal(p[:, :, :30, :50], img * (1 / 255.0)) and p[:, :, 30:].abs().max() == 0 and p[:, :, :, 50:].abs().max() == 0
    assert torch.equal(ops.crop_nchw(img.cuda(), 11, 13).cpu(), img[:, :, :11, :13])
    # depthwise dilated 3x3 + folded BN + ReLU
    C = 64
    xd = _q(_gen((2, C, 13, 17), 3), dtype)
    w = _gen((C, 1, 3, 3), 4) / 3
    s, t = 1 + 0.1 * _gen((C,), 5), 0.1 * _gen((C,), 6)
    for dil in (1, 6, 12, 18):
        ref = F.relu(F.conv2d(xd, w, None, padding=dil, dilation=dil, groups=C) * s[None, :, None, None] + t[None, :, None, None])
        got = ops.dwconv3x3(to_tokens(xd, dtype), w.reshape(C, 9).t().contiguous().cuda(), s.cuda(), t.cuda(), 2, 13, 17, dil)
        assert rel_err(from_tokens(got, 2, 13, 17), ref)[0] < (1e-6 if dtype == torch.float32 else 8e-3), dil
    # wide maps take the comb kernel (4 outputs per thread, one dilation apart): ragged width, every dilation, column window
    xw = _q(_gen((2, C, 21, 77), 13), dtype)
    for dil in (1, 6, 12, 18):
        ref = F.relu(F.conv2d(xw, w, None, padding=dil, dilation=dil, groups=C) * s[None, :, None, None] + t[None, :, None, None])
        buf = torch.zeros((2 * 21 * 77, 3 * C), dtype=dtype, device="cuda")
        ops.dwconv3x3(to_tokens(xw, dtype), w.reshape(C, 9).t().contiguous().cuda(), s.cuda(), t.cuda(), 2, 21, 77, dil,
                      out=buf[:, C:2 * C])
        assert rel_err(from_tokens(buf[:, C:2 * C], 2, 21, 77), ref)[0] < (1e-6 if dtype == torch.float32 else 8e-3), dil
        assert buf[:, :C].abs().max().item() == 0 and buf[:, 2 * C:].abs().max().item() == 0
    # the comb kernel with per-XCD channel slabs and the lattice kernel (large tensors; forced here): all variants accumulate in the same
    # order -> identical bits; ragged sizes, column window, every dilation
    import os
    C2 = 128
    xc = _q(_gen((2, C2, 35, 83), 14), dtype)
    w2 = _gen((C2, 1, 3, 3), 15) / 3
    s2, t2 = 1 + 0.1 * _gen((C2,), 16), 0.1 * _gen((C2,), 17)
    for dil in (1, 6, 12, 18):
        ref = F.relu(F.conv2d(xc, w2, None, padding=dil, dilation=dil, groups=C2) * s2[None, :, None, None] + t2[None, :, None, None])
        outs = []
        for kern in ("1", "2", "3", "4"):   # 4 = the lattice kernel (one residue class of the dilation per workgroup)
            os.environ["MADM_DWCONV_KERNEL"] = kern
            try:
                buf = torch.zeros((2 * 35 * 83, 3 * C2), dtype=dtype, device="cuda")
                ops.dwconv3x3(to_tokens(xc, dtype), w2.reshape(C2, 9).t().contiguous().cuda(), s2.cuda(), t2.cuda(), 2, 35, 83,
                              dil, out=buf[:, C2:2 * C2])
            finally:
                del os.environ["MADM_DWCONV_KERNEL"]
            assert rel_err(from_tokens(buf[:, C2:2 * C2], 2, 35, 83), ref)[0] < (1e-6 if dtype == torch.float32 else 8e-3), (dil, kern)
            assert buf[:, :C2].abs().max().item() == 0 and buf[:, 2 * C2:].abs().max().item() == 0
            outs.append(buf)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[3]), dil
    # tanh gates (prompt / time conditioning) with the batch repeat
    a1, x1, a2, x2 = torch.rand(1, 77, 768), _gen((1, 77, 768), 7), torch.rand(1, 77, 768), _gen((1, 77, 768), 8)
    g = ops.tanh_gate(x1.cuda(), a1.cuda(), x2.cuda(), a2.cuda(), repeat=3).cpu()
    assert g.shape == (3, 77, 768) and rel_err(g, (torch.tanh(a1) * x1 + torch.tanh(a2) * x2).repeat(3, 1, 1))[0] < 1e-6
    # argmax: first maximal channel wins (bit-exact index op)
    lg = _gen((2, 11, 9, 10), 9)
    lg[0, 3, 2, 2] = lg[0, 7, 2, 2] = 100.0
    lab = ops.argmax_nchw(lg.cuda()).cpu()
    assert torch.equal(lab, lg.argmax(dim=1)) and lab[0, 2, 2].item() == 3


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(320, 512, 16, 16), (3, 128, 16, 32), (512, 512, 8, 16)], ids=["s3", "s0_rgb", "same_width"])
def test_bottleneck_block(cuda, dtype, case):
    """detectron2 BottleneckBlock(GN) of the feature projections (feature_extractor.py:347-359)."""
    from madm_amd import weights
    from madm_amd.backbone import BottleneckBlock
    from madm_amd.nn import Tok
    from oracle import third_party as tp
    cin, cout, H, W = case
    blk = weights.synth_init_(BottleneckBlock(cin, cout, bottleneck_channels=128), 5, "p.").cuda()
    ref_blk = weights.synth_init_(tp.BottleneckBlock(cin, cout, bottleneck_channels=128, norm="GN"), 5, "p.").eval()
    assert set(blk.state_dict()) == set(ref_blk.state_dict())
    x = _q(_gen((2, cin, H, W), 1), dtype)
    with torch.no_grad():
        ref = ref_blk(x)
    cpad = cin if cin % 64 == 0 else 4
    got = blk(Tok(to_tokens(x, dtype, cpad), 2, H, W))
    e, l2 = rel_err(from_tokens(got.t, 2, H, W), ref)
    assert e < (3e-5 if dtype == torch.float32 else 3e-2), f"{e:.2e} {l2:.2e}"


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_daformer_head_small(cuda, dtype):
    """DAFormerHead (MLP embeds -> bilinear -> concat -> sep-ASPP -> 3x3 bottleneck -> 1x1 classes) against the oracle
    restatement on mmcv-shaped modules, eval mode, 3 scales."""
    from madm_amd import weights
    from madm_amd.head import DAFormerHead
    from madm_amd.backbone import FeatureDict
    from madm_amd.nn import Tok
    from oracle import madm_path
    cfg = dict(madm_path.S345_CFG)
    head = DAFormerHead(in_channels=cfg["head_in_channels"], in_keys=cfg["out_features"], in_index=[0, 1, 2], channels=256,
                        num_classes=11, norm_cfg=dict(type='BN'), decoder_params=madm_path.head_decoder_params())
    ref = madm_path.OracleHead(cfg).eval()
    for m in (head, ref):
        weights.synth_init_(m, 2, "head.")
        weights.synth_buffers_(m, 2, "head.")
    assert set(head.state_dict()) == set(ref.state_dict())
    head = head.cuda().eval()
    feats = {"s3": _q(_gen((2, 512, 16, 32), 1), dtype), "s4": _q(_gen((2, 512, 8, 16), 2), dtype),
             "s5": _q(_gen((2, 512, 4, 8), 3), dtype)}
    with torch.no_grad():
        want = ref({'output_features': feats})
    fd = FeatureDict()
    fd.tok = {k: Tok(to_tokens(v, dtype), 2, v.shape[2], v.shape[3]) for k, v in feats.items()}
    for k in feats:
        fd[k] = None
    got = head({'output_features': fd}).cpu()
    e, l2 = rel_err(got, want)
    assert got.shape == want.shape and e < (3e-5 if dtype == torch.float32 else 3e-2), f"{e:.2e} {l2:.2e}"


def _build_product(cfg_name, dtype, lora_configs=()):
    from madm_amd.ldm_rocm import LdmRocm
    from madm_amd.backbone import BasePromptTimeGenerator, AttentionFeatureExtractorBackbone
    from madm_amd.head import DAFormerHead
    from madm_amd.meta_arch import MadmInference
    from oracle import madm_path
    cfg = madm_path.cfg_by_name(cfg_name)
    ldm = LdmRocm("", encoder_block_indices=[], unet_block_indices=[5, 8, 11], decoder_block_indices=(),
                  input_range='-1+1', unet_block_indices_type='after', finetune_unet='no', compute_dtype=dtype,
                  weights='synthetic', seed=0, vae_decoder_loss=cfg["vae_decoder_loss"])
    gen = BasePromptTimeGenerator(learnable_cond_prompt=True, learnable_cond_time=True, clip_state='no', num_timesteps=1,
                                  clip_model_name="ViT-L-14-336", ldm_extractor=ldm, same_cond_params=True)
    backbone = AttentionFeatureExtractorBackbone(
        attention_features_res=None, feature_dims=list(cfg["feature_dims"]), projection_dim=list(cfg["projection_dim"]),
        attention_features_location=None, feature_extractor=gen, num_res_blocks=1, out_features=list(cfg["out_features"]))
    n = len(cfg["out_features"])
    head = DAFormerHead(in_channels=list(cfg["head_in_channels"]), in_keys=list(cfg["out_features"]), in_index=list(range(n)),
                        channels=256, dropout_ratio=0.1, num_classes=cfg["num_classes"], norm_cfg=dict(type='BN'),
                        align_corners=False, decoder_params=madm_path.head_decoder_params())
    init_eval_params(backbone, head)
    model = MadmInference(backbone.cuda(), head.cuda(), target_modality="Depth", lora_configs=lora_configs).eval()
    if lora_configs:
        seed_lora_(ldm.unet)
    return model


@pytest.mark.parametrize("name", ["eval_s345", "eval_depth", "eval_infrared", "eval_depth_lora"])
@pytest.mark.parametrize("dtype", DTYPES + [torch.float16], ids=["f32", "bf16", "f16"])
def test_eval_forward_golden(cuda, name, dtype):
    """MTMADISE eval forward (mtmadise.py:657-691) end to end vs the vectors of the reference's own classes."""
    from madm_amd.meta_arch import MadmInference
    case = EVAL_CASES[name]
    gold = load_golden(name)
    model = _build_product(case["cfg"], dtype, lora_configs=case.get("lora_configs", ()))
    img = eval_image(case["H"], case["W"])
    unet = model.backbone.feature_extractor.ldm_extractor.unet
    if case.get("lora_configs"):      # the 'name_rN_aM' contract (mtmadise.py:48-54,115-127): parsed, all adapters active
        assert model.lora_configs == {'default': dict(rank=8, alpha=8), 'Depth': dict(rank=8, alpha=16)}
        assert model.active_lora_adapter() == ['default', 'Depth']
        wrapped = [n for n, m in unet.named_modules() if hasattr(m, "_active_adapter")]
        assert len(wrapped) == 16 * 2 * 4 and all(n.endswith(("to_q", "to_k", "to_v", "to_out.0")) for n in wrapped)
        assert sum(p.numel() for n, p in unet.named_parameters() if ".lora_" in n) == 2 * 199296 * 8
        assert not any(p.requires_grad for p in unet.parameters())       # finetune_unet='no': _freeze() froze the adapters too
    out = model([{"target_second_modality": img}])
    torch.cuda.synchronize()
    if case.get("lora_configs"):      # :672 selected the target modality's adapter
        assert model.active_lora_adapter() == ['Depth']
        base = load_golden("eval_depth")["sem_seg"]
        assert rel_err(out[0]["sem_seg"].cpu()[:, :, ::4, ::4], base)[1] > 0.02, "the adapter must change the result"
    sem = out[0]["sem_seg"].cpu()
    assert tuple(sem.shape) == tuple(gold["sem_seg_shape"].tolist())
    e, l2 = rel_err(sem[:, :, ::4, ::4], gold["sem_seg"])
    labels = MadmInference.predict_labels(out[0]).cpu()
    assert torch.equal(labels, sem[0].argmax(dim=0))          # the device argmax is the torch index op, bit for bit
    gl = gold["labels"].long()
    margin = gold["top2_margin"].float()
    scale = gold["sem_seg"].abs().max().item()
    agree = (labels == gl).float().mean().item()
    print(name, dtype, f"sem_seg max {e:.2e} l2 {l2:.2e}; label agreement {agree:.5f}")
    if dtype == torch.float32:
        assert e < 1e-3
        decided = margin > 1e-3 * scale                       # pixels whose top-2 logits are not within fp32 noise
        assert torch.equal(labels[decided], gl[decided]), "labels differ where the reference margin is decisive"
        assert agree > 0.9999
    elif dtype == torch.float16:
        assert l2 < 2e-2 and agree > 0.995
    else:
        assert l2 < 8e-2 and agree > 0.97
    feats = model.backbone(model_input(model, img), input_modal='others')['output_features']
    for k in feats:
        g = gold["feat_" + k]
        f = feats[k].cpu()
        assert tuple(f.shape) == tuple(gold["feat_" + k + "_shape"].tolist())
        st = max(1, f.shape[-1] // 32)
        ef, lf = rel_err(f[:, ::8, ::st, ::st], g)
        assert (ef < 1e-3) if dtype == torch.float32 else (lf < (2e-2 if dtype == torch.float16 else 8e-2)), (k, ef, lf)


def model_input(model, img):
    from madm_amd import ops
    H, W = img.shape[1:]
    Hp, Wp = (H + 63) // 64 * 64, (W + 63) // 64 * 64
    return ops.scale_pad_nchw(img[None].cuda().contiguous(), 1 / 255.0, Hp, Wp)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_sliding_window_backbone(cuda, dtype):
    """slide_inference (feature_extractor.py:199-278): three 512-wide windows of a 512 x 1024 image as one batched forward,
    features averaged by window count.  Vector from the oracle restatement (the reference's slide_forward raises a
    TypeError on its own Attention backbone, see tests/golden/gen_golden.py::main_slide)."""
    from madm_amd.ldm_rocm import LdmRocm
    from madm_amd.backbone import BasePromptTimeGenerator, AttentionFeatureExtractorBackbone
    from madm_amd.head import DAFormerHead
    from oracle import madm_path
    from madm_amd import ops
    cfg = madm_path.S345_CFG
    ldm = LdmRocm("", [], [5, 8, 11], (), input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                  compute_dtype=dtype, weights='synthetic', seed=0)
    gen = BasePromptTimeGenerator(ldm_extractor=ldm, same_cond_params=True)
    backbone = AttentionFeatureExtractorBackbone(None, list(cfg["feature_dims"]), None, feature_extractor=gen,
                                                 out_features=list(cfg["out_features"]),
                                                 projection_dim=list(cfg["projection_dim"]), slide_inference=True)
    head = DAFormerHead(in_channels=[512] * 3, in_keys=cfg["out_features"], in_index=[0, 1, 2], channels=256, num_classes=11,
                        decoder_params=madm_path.head_decoder_params())
    init_eval_params(backbone, head)
    backbone = backbone.cuda().eval()
    img = torch.rand((1, 3, 512, 1024), generator=torch.Generator().manual_seed(778))
    feats = backbone(img.cuda(), input_modal='others')['output_features']
    gold = load_golden("slide_s345")
    for k in ("s3", "s4", "s5"):
        f = feats[k].cpu()
        assert tuple(f.shape) == tuple(gold["feat_" + k + "_shape"].tolist())
        e, l2 = rel_err(f[:, ::8], gold["feat_" + k])
        assert (e < 1e-3) if dtype == torch.float32 else (l2 < 6e-2), (k, e, l2)
    # the merge kernel on its own: counts 1 / 2 / 2 / 1 over the four 256-column bands at stride 8
    w = torch.randn(3 * 2 * 4 * 64, 64)
    m = ops.slide_merge(w.to(dtype).cuda(), 3, 2, 4, 64, 128, [0, 32, 64]).float().cpu().reshape(2, 4, 128, 64)
    wq = w.to(dtype).float().reshape(3, 2, 4, 64, 64)
    ref = torch.zeros(2, 4, 128, 64); cnt = torch.zeros(1, 1, 128, 1)
    for k, x1 in enumerate([0, 32, 64]):
        ref[:, :, x1:x1 + 64] += wq[k]; cnt[:, :, x1:x1 + 64] += 1
    assert rel_err(m, ref / cnt)[0] < (1e-6 if dtype == torch.float32 else 8e-3)


def test_device_evaluator_is_bit_exact(cuda):
    """argmax -> confusion matrix -> mIoU on the device against the numpy restatement of
    evaluation/d2_evaluator.py:106-127,246-270 (integer work: bit-exact; metrics: float64-equal)."""
    import numpy as np
    from madm_amd.evaluation import SemSegEvaluator
    from oracle import madm_path
    K = 11
    g = torch.Generator().manual_seed(5)
    ev = SemSegEvaluator(K, ignore_label=255)
    conf_ref = np.zeros((K + 1, K + 1), dtype=np.int64)
    for i in range(3):
        logits = torch.randn((1, K, 97, 131), generator=g)
        logits[0, 2, 5, 5] = logits[0, 9, 5, 5] = 50.0                      # tie -> first maximal class
        gt = torch.randint(0, K, (1, 97, 131), generator=g)
        gt[0, :7] = 255                                                        # ignored rows
        ev.process([{"target_label": gt}], [{"sem_seg": logits.cuda()}])
        conf_ref += madm_path.evaluator_confusion(logits[0].argmax(dim=0).numpy(), gt[0].numpy(), K, 255)
    assert np.array_equal(ev.confusion(), conf_ref) and conf_ref.sum() == 3 * 97 * 131
    res = ev.evaluate()["sem_seg"]
    ref = madm_path.evaluator_metrics(conf_ref, K)
    for k in ("mIoU", "fwIoU", "mACC", "pACC"):
        assert res[k] == ref[k], k
    assert all(res[f"IoU-{i}"] == 100 * ref["iou"][i] for i in range(K))


def test_inference_on_dataset_pipelined_matches_serial_loop(cuda):
    """evaluation.inference_on_dataset (the reference's loop, evaluator.py:75-93, on whole-forward hipGraphs with four images
    in flight and evaluator.process chained on the slot's stream): eight DIFFERENT images of two sizes -> every sem_seg
    bit-identical to MadmInference.forward on that image, the confusion matrix and the metrics equal to the serial loop's;
    an out-of-range image (values > 255) raises the reference's range assert, late."""
    import numpy as np
    from madm_amd.evaluation import SemSegEvaluator, inference_on_dataset
    model = _build_product("DEPTH", torch.float16)
    K = 11
    g = torch.Generator().manual_seed(31)
    loader = []
    for i in range(8):
        H, W = (512, 512) if i not in (2, 5) else (448, 512)
        img = 255.0 * torch.rand((3, H, W), generator=g)
        if i % 2:
            img = img.to(torch.uint8)                                   # dataset mappers hand uint8 or float images over
        loader.append([{"target_second_modality": img.cuda() if i % 3 else img,
                        "target_label": torch.randint(0, K, (1, H, W), generator=g)}])

    class Recording(SemSegEvaluator):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.seen = []

        def process(self, inputs, outputs):
            self.seen.append(outputs[0]["sem_seg"].clone())             # on the slot's stream, behind its forward
            super().process(inputs, outputs)

    ev = Recording(K, ignore_label=255)
    res = inference_on_dataset(model, loader, ev)
    ev2 = SemSegEvaluator(K, ignore_label=255)
    from madm_amd import ops
    for i, inputs in enumerate(loader):
        with ops.tuning_profile("throughput", pin=True):     # the rows the runner's graphs were captured under (bit-for-bit comparison)
            out = model(inputs)
        assert torch.equal(out[0]["sem_seg"], ev.seen[i]), f"image {i}: pipelined forward differs from forward()"
        ev2.process(inputs, out)
    assert np.array_equal(ev.confusion(), ev2.confusion()) and ev.confusion().sum() == 6 * 512 * 512 + 2 * 448 * 512
    assert res["sem_seg"]["mIoU"] == ev2.evaluate()["sem_seg"]["mIoU"]
    bad = [{"target_second_modality": loader[0][0]["target_second_modality"] * 1.01 + 1.0,
            "target_label": loader[0][0]["target_label"]}]
    with pytest.raises(AssertionError, match="input range check"):
        inference_on_dataset(model, [loader[0], bad, loader[4], loader[6]], SemSegEvaluator(K, ignore_label=255))


def test_staged_inference_matches_forward(cuda):
    """pipeline.StagedInference (round 6: VAE encoder | UNet | decoder + projections + head as three stage graphs per image, the
    boundaries taken by LdmRocm.stage_hook inside one ordinary model call): ten DIFFERENT images through six slots, up to six
    in flight -> every sem_seg bit-identical to MadmInference.forward on that image; the evaluator chained on the last
    stage's stream sees the same confusion matrix as the serial loop; the deferred range assert still fires."""
    import numpy as np
    from madm_amd.evaluation import SemSegEvaluator, inference_on_dataset
    from madm_amd.pipeline import StagedInference
    model = _build_product("DEPTH", torch.float16)
    K = 11
    g = torch.Generator().manual_seed(41)
    imgs = [255.0 * torch.rand((3, 512, 512), generator=g) for _ in range(10)]
    calls = [[{"target_second_modality": (im.to(torch.uint8) if i % 2 else im).cuda(),
               "target_label": torch.randint(0, K, (1, 512, 512), generator=g)}] for i, im in enumerate(imgs)]
    runner = StagedInference(model, calls[0], unet_streams=2, slots=6)
    got = []
    for c in calls:
        out, done, slot = runner.submit(c)
        with torch.cuda.stream(runner.stream_of(slot)):
            got.append(out[0]["sem_seg"].clone())
    runner.drain()
    from madm_amd import ops
    with ops.tuning_profile("throughput", pin=True):         # the rows the stage graphs were captured under
        for i, c in enumerate(calls):
            want = model(c)[0]["sem_seg"]
            assert torch.equal(got[i], want), f"image {i}: staged forward differs from forward()"
    ev, ev2 = SemSegEvaluator(K, ignore_label=255), SemSegEvaluator(K, ignore_label=255)
    res = inference_on_dataset(model, calls, ev, runner="staged")
    with ops.tuning_profile("throughput", pin=True):
        for c in calls:
            ev2.process(c, model(c))
    assert np.array_equal(ev.confusion(), ev2.confusion()) and res["sem_seg"]["mIoU"] == ev2.evaluate()["sem_seg"]["mIoU"]
    bad = [{"target_second_modality": calls[0][0]["target_second_modality"] * 1.01 + 1.0, "target_label": calls[0][0]["target_label"]}]
    with pytest.raises(AssertionError, match="input range check"):
        inference_on_dataset(model, [calls[0], bad, calls[2], calls[4]], SemSegEvaluator(K, ignore_label=255), runner="staged")
    assert model.backbone.feature_extractor.ldm_extractor.__dict__.get("stage_hook") is None


def test_flat_adamw_ema_clip(cuda):
    """One-launch AdamW (+ folded unscale / clip) and EMA on flat fp32 storage against torch.optim.AdamW,
    clip_grad_norm_ and the reference's EMA formula (cmdise.py:337-349) on CPU."""
    import copy
    from madm_amd.optim import FlatParams, FlatAdamW, ema_update, grad_sumsq
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 11))
    ref = copy.deepcopy(net)
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=5e-3, weight_decay=0.05)
    dev = copy.deepcopy(net).cuda()
    flat = FlatParams(list(dev.parameters()))
    opt = FlatAdamW(flat, lr=5e-3, weight_decay=0.05)
    ema_ref = [p.detach().clone() for p in ref.parameters()]
    ema_flat = flat.flat.clone()
    for it in range(4):
        grads = [torch.randn(p.shape, generator=torch.Generator().manual_seed(10 * it + i)) * (3.0 if it % 2 else 0.01)
                 for i, p in enumerate(ref.parameters())]
        loss_scale = 128.0
        for p, pd, g in zip(ref.parameters(), dev.parameters(), grads):
            p.grad = g.clone()
            pd.grad.copy_((g * loss_scale).cuda())                 # scaled grads, as GradScaler leaves them
        n_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5).item()
        opt_ref.step()
        n_dev = opt.step(clip_grad=0.5, loss_scale=loss_scale)
        assert abs(n_dev - n_ref) / n_ref < 1e-5
        alpha = min(1 - 1 / (it + 1), 0.999)
        ema_ref = [alpha * e + (1 - alpha) * p.detach() for e, p in zip(ema_ref, ref.parameters())]
        ema_update(ema_flat, flat.flat, alpha)
    torch.cuda.synchronize()
    for p, pd in zip(ref.parameters(), dev.parameters()):
        assert rel_err(pd.detach().cpu(), p.detach())[0] < 2e-6
    off = 0
    for e, p in zip(ema_ref, dev.parameters()):
        got = ema_flat[off:off + p.numel()].view_as(p).cpu()
        assert rel_err(got, e)[0] < 2e-6
        off += (p.numel() + 3) // 4 * 4
    x = torch.randn(100003)
    assert abs(grad_sumsq(x.cuda()[:100000].contiguous()).item() - (x[:100000].double() ** 2).sum().item()) < 1e-6 * 1e5
