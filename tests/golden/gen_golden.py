"""Generates the committed golden vectors of the hot path.  RUNS ONLY IN THE BUILD CONTAINER (needs
/root/reference): the reference's OWN functions ``vae_encoder`` / ``add_noise`` / ``diffusion_unet``
(modeling/meta_arch/ldm_diffusers.py:283-311,349-360,454-616), loaded by path, drive the CPU fp32
oracle modules (oracle/sd_modules.py) filled with seeded synthetic parameters (madm_amd/weights.py).
Inputs are regenerated from seeds by the tests (``make_inputs``); only outputs are stored.

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import sd_modules, ref_driver  # noqa: E402
from madm_amd import weights  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
WEIGHT_SEED = 0

sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import CASES, CH_STRIDE_FULL, make_inputs, add_lora, EVAL_CASES, eval_image, init_eval_params  # noqa: E402


def build_oracle(lora=False, seed=WEIGHT_SEED):
    vae = sd_modules.AutoencoderKL()
    unet = sd_modules.UNet2DConditionModel()
    weights.synth_init_(vae, seed, "vae.")
    weights.synth_init_(unet, seed, "unet.")
    if lora:
        add_lora(unet, sd_modules.LoraConfig, seed)
    return vae.eval(), unet.eval()


def run_reference(ref, vae, unet, sched, case):
    images, cond_inputs, cond_emb, timesteps, shared_noise = make_inputs(**case)
    with torch.no_grad():
        x = (images - 0.5) / 0.5   # LdmDiffusers.forward :145-146
        latents, _ = ref.vae_encoder(vae=vae, images=x, encoder_block_indices=[])
        noisy = ref.add_noise(noise_scheduler=sched, latents=latents, timesteps=timesteps, shared_noise=shared_noise)
        out, feats = ref.diffusion_unet(unet=unet, sample=noisy, timestep=timesteps, encoder_hidden_states=cond_inputs,
                                        res_time_embedding=cond_emb.clone(), unet_block_indices=[5, 8, 11],
                                        unet_block_indices_type='after')
        dec, _ = ref.vae_decoder(vae=vae, latents=out.sample, decoder_block_indices=[], output_final=True)  # :194
    return latents, noisy, out.sample, feats, dec


def main():
    only = sys.argv[1:]
    assert ref_driver.available(), "needs /root/reference"
    ref = ref_driver.load()
    sched = sd_modules.DDPMScheduler()
    torch.set_num_threads(os.cpu_count())
    models = {}
    for name, case in CASES.items():
        if only and name not in only:
            continue
        key = case["lora"]
        if key not in models:
            models[key] = build_oracle(lora=key)
        vae, unet = models[key]
        t0 = time.time()
        latents, noisy, sample, feats, dec = run_reference(ref, vae, unet, sched, case)
        stride = CH_STRIDE_FULL if name.startswith("full") else 1
        dstride = 4 if name.startswith("full") else 1   # the 512x512 decoder image is stored 4x subsampled
        out = {"latents": latents.numpy(), "noisy": noisy.numpy(), "sample": sample.numpy(),
               "decoder": dec[:, :, ::dstride, ::dstride].contiguous().numpy(),
               "decoder_shape": np.array(dec.shape, dtype=np.int64)}
        for i, f in enumerate(feats):
            out[f"tap{i}"] = f[:, ::stride].contiguous().numpy()
            out[f"tap{i}_stats"] = np.array([f.mean().item(), f.std().item(), f.abs().max().item()], dtype=np.float64)
            out[f"tap{i}_shape"] = np.array(f.shape, dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(f"{name}: {time.time() - t0:.1f}s", {k: v.shape for k, v in out.items() if not k.endswith('_stats')})


def main_eval():
    """Full inference forward through the REFERENCE's BasePromptTimeGenerator / AttentionFeatureExtractorBackbone /
    DAFormerHead classes (loaded by path) on the oracle modules; eval post-processing per mtmadise.py:657-691."""
    from oracle import madm_path
    only = sys.argv[1:]
    ns = ref_driver.load_modeling()
    vae, unet = build_oracle(lora=False)
    torch.set_num_threads(os.cpu_count())
    for name, case in EVAL_CASES.items():
        if only and name not in only:
            continue
        cfg = madm_path.cfg_by_name(case["cfg"])
        if case.get("lora_configs"):      # own modules: set_multi_lora wraps the attention projections in place
            vae_c, unet_c = build_oracle(lora=False)
        else:
            vae_c, unet_c = vae, unet
        backbone, head = madm_path.build_reference_eval_model(ns, vae_c, unet_c, cfg)
        init_eval_params(backbone, head)
        t0 = time.time()
        if case.get("lora_configs"):
            # the meta-architecture's eval branch (mtmadise.py:657-691) incl. the LoRA contract of :48-54,115-147,672 as
            # oracle/train_path.OracleMTMADISE restates it, over the REFERENCE's backbone / head classes
            from oracle import train_path
            from golden_util import seed_lora_
            meta = train_path.OracleMTMADISE(backbone, head, None, target_modality='Depth', train_palette=[0, 0, 0],
                                             lora_configs=case["lora_configs"], init_ema=False).eval()
            seed_lora_(unet_c)
            sem_seg = meta.forward_eval([{'target_second_modality': eval_image(case["H"], case["W"])}])[0]['sem_seg']
            feats = meta.last_eval_feats
            assert all(m._active_adapter == ['Depth'] for m in unet_c.modules() if hasattr(m, '_active_adapter'))
        else:
            sem_seg, feats = madm_path.eval_forward(backbone, head, eval_image(case["H"], case["W"]))
        out = {"sem_seg": sem_seg[:, :, ::4, ::4].contiguous().numpy(),
               "sem_seg_shape": np.array(sem_seg.shape, dtype=np.int64),
               "labels": sem_seg[0].argmax(dim=0).to(torch.uint8).numpy(),
               "top2_margin": (lambda t: (t[0] - t[1]))(sem_seg[0].topk(2, dim=0).values).to(torch.float16).numpy()}
        for k, f in feats['output_features'].items():
            st = max(1, f.shape[-1] // 32)
            out["feat_" + k] = f[:, ::8, ::st, ::st].contiguous().numpy()
            out["feat_" + k + "_shape"] = np.array(f.shape, dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(f"{name}: {time.time() - t0:.1f}s", {k: v.shape for k, v in out.items()})


def main_slide():
    """Sliding-window inference (feature_extractor.py:199-278), 1 x 3 x 512 x 1024 input, s3/s4/s5 features.
    NOTE: the reference's own slide_forward cannot run on its AttentionFeatureExtractorBackbone -- `channel =
    self._out_feature_channels[k]` is the projection_dim LIST there (feature_extractor.py:111,208-214 ->
    "TypeError: zeros(): argument 'size' ... got list", reproduced in this container) -- so this vector comes from the
    oracle's restatement of the documented algorithm (oracle/madm_path.slide_forward), not from the reference class."""
    from oracle import madm_path
    vae, unet = build_oracle(lora=False)
    torch.set_num_threads(os.cpu_count())
    cfg = madm_path.S345_CFG
    backbone, head = madm_path.build_oracle_eval_model(vae, unet, cfg)
    init_eval_params(backbone, head)
    img = torch.rand((1, 3, 512, 1024), generator=torch.Generator().manual_seed(778))
    t0 = time.time()
    with torch.no_grad():
        feats = madm_path.slide_forward(backbone, img)['output_features']
    out = {}
    for k, f in feats.items():
        out["feat_" + k] = f[:, ::8].contiguous().numpy()
        out["feat_" + k + "_shape"] = np.array(f.shape, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "slide_s345.npz"), **out)
    print(f"slide_s345: {time.time() - t0:.1f}s", {k: v.shape for k, v in out.items()})


def build_train_oracle(reference=True, variant="train_depth"):
    """OracleMTMADISE of the Depth config at TRAIN_CASE size; ``reference``: the REFERENCE's DAFormerHead class and
    CmdiseCriterion (this container), else the oracle's restatements (anywhere)."""
    from oracle import madm_path, train_path
    from golden_util import TRAIN_CASE, TRAIN_VARIANTS, train_palette, train_dropout_scales, model_args, prepare_lora_
    cfg = madm_path.DEPTH_CFG
    vae, unet = build_oracle(lora=False)
    crit = train_path.reference_criterion() if reference else None
    model = train_path.build(vae, unet, cfg, criterion_cls=crit, in_size=TRAIN_CASE["size"],
                             train_palette=train_palette(TRAIN_CASE["K"]), pseudo_threshold=TRAIN_CASE["pseudo_threshold"],
                             finetune_unet=TRAIN_VARIANTS[variant].get("finetune_unet", "all"), **model_args(variant))
    prepare_lora_(unet, variant)
    if reference:
        ns = ref_driver.load_modeling()
        n = len(cfg["out_features"])
        head = ns.daformer_head.DAFormerHead(
            in_channels=list(cfg["head_in_channels"]), in_keys=list(cfg["out_features"]), in_index=list(range(n)),
            channels=256, dropout_ratio=0.1, num_classes=cfg["num_classes"], norm_cfg=dict(type='BN', requires_grad=True),
            align_corners=False, decoder_params=madm_path.head_decoder_params())
        head.dropout = train_path.FixedDropout2d()
        model.sem_seg_head = model.sem_seg_head_sec_modal = head
    init_eval_params(model.backbone, model.sem_seg_head)
    model._inti_ema_weights()           # teacher = copy of the initialised student (cmdise.py:307-335)
    model.train()
    model.backbone.feature_extractor.ldm_extractor.vae.eval()
    model.backbone.feature_extractor.ldm_extractor.unet.eval()                   # LdmDiffusers._freeze: train(mode=False)
    sc = train_dropout_scales(TRAIN_CASE["B"])
    model.sem_seg_head.dropout.scales = [sc[0], sc[1]]
    model.ema_sem_seg_head.dropout.scales = [sc[2]]
    return model


def run_train_step(model):
    import random
    from golden_util import TRAIN_CASE, train_inputs, grad_summary
    random.seed(TRAIN_CASE["py_seed"])
    np.random.seed(TRAIN_CASE["np_seed"])
    losses = model.forward_train(train_inputs(**TRAIN_CASE))
    total = sum(losses.values())
    total.backward()
    named = [(n, p.grad) for n, p in model.named_parameters() if p.requires_grad and p.grad is not None]
    names, rows, full = grad_summary(named, TRAIN_CASE["full_grad_max_numel"])
    out = {"loss_" + k: np.array(v.item(), dtype=np.float64) for k, v in losses.items()}
    out["grad_names"] = np.array("\n".join(names))
    out["grad_rows"] = rows
    for n, g in full.items():
        out["grad:" + n] = g.numpy()
    ls = model.last_step
    out["mixed_lbl"] = ls["mixed_lbl"].to(torch.uint8).numpy()
    out["pseudo_label"] = ls["pseudo_label"].to(torch.uint8).numpy()
    out["pseudo_weight0"] = np.array(ls["pseudo_weight"].flatten()[0].item())
    out["mixed_seg_weight"] = ls["mixed_seg_weight"].numpy()
    out["source_logits"] = ls["source_logits"].detach().numpy()
    out["target_logits"] = ls["target_logits"].detach().numpy()
    out["ema_logits"] = ls["ema_logits"].detach().numpy()
    for tag, head in (("student", model.sem_seg_head), ("teacher", model.ema_sem_seg_head)):
        for n, b in head.named_buffers():
            if n.endswith("running_mean") or n.endswith("running_var"):
                out[f"bn:{tag}:{n}"] = b.numpy().copy()
    return out


def main_train():
    """One training step (mtmadise.py:180-656, shipped Depth flags, 64 x 64) through oracle/train_path.OracleMTMADISE with
    the REFERENCE's DAFormerHead and CmdiseCriterion; losses, gradient checksums of every trainable tensor, full
    gradients of the small non-UNet tensors, pseudo labels and BatchNorm running statistics."""
    from golden_util import TRAIN_VARIANTS
    torch.set_num_threads(os.cpu_count())
    for variant in TRAIN_VARIANTS:
        if sys.argv[1:] and variant not in sys.argv[1:]:
            continue
        t0 = time.time()
        out = run_train_step(build_train_oracle(reference=True, variant=variant))
        np.savez_compressed(os.path.join(HERE, variant + ".npz"), **out)
        print(f"{variant}: {time.time() - t0:.1f}s", {k: float(v) for k, v in out.items() if k.startswith("loss_")},
              len(out["grad_rows"]), "gradient tensors")


def main_labels():
    """Label pipeline through the REFERENCE's own functions (extracted by AST from mtmadise.py / dacs_transforms.py,
    oracle/labels.reference_functions): palette conversion, ClassMix masks with a seeded numpy RNG, one_mix; the
    pseudo-label block (mtmadise.py:339-349) is plain torch and is restated in oracle/labels.pseudo_labels."""
    from oracle import labels as L
    from golden_util import LABEL_CASE, label_inputs
    conv, gcm, _, mix = L.reference_functions()
    inp = label_inputs(**LABEL_CASE)
    pal768 = inp["palette"] + [0] * (768 - len(inp["palette"]))
    rgb, valid = conv(inp["label"].clone(), pal768)
    np.random.seed(LABEL_CASE["mix_seed"])
    masks = gcm(inp["label"])
    mixed_img, mixed_lbl = mix(masks[0], data=torch.stack((inp["imgs"][0], inp["imgs"][1])),
                               target=torch.stack((inp["label"][0], inp["label"][1])))
    prob, plabel, pweight = L.pseudo_labels(inp["logits"], (LABEL_CASE["H"], LABEL_CASE["W"]), LABEL_CASE["thr"])
    out = dict(rgb=rgb.numpy(), valid=valid.numpy(), mask0=masks[0].numpy(), mask1=masks[1].numpy(),
               mixed_img=mixed_img.numpy(), mixed_lbl=mixed_lbl.numpy(), prob=prob.numpy(), plabel=plabel.numpy(),
               pweight=np.float32(pweight.flatten()[0].item()))
    np.savez_compressed(os.path.join(HERE, "labels.npz"), **out)
    print("labels.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    args = sys.argv[1:]
    if not args or any(a.startswith("train_") for a in args):
        main_train()
    if not args or "labels" in args:
        main_labels()
    if not args or "slide_s345" in args:
        main_slide()
    if not args or any(a in CASES for a in args):
        main()
    if not args or any(a in EVAL_CASES for a in args):
        main_eval()
