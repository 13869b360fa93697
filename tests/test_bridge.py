"""On-disk / checkpoint bridge of the path: CPU-only tests (no kernel launches).

* ``weights.load_diffusers_dir``: a synthetic CompVis/stable-diffusion-v1-4-shaped snapshot (safetensors, diffusers 0.25
  names; the VAE attention block in the pre-0.25 ``query/key/value/proj_attn`` conv-shaped form that the published
  checkpoint uses) loads into the product parameter trees bit for bit;
* ``state_dict()`` keys of the product meta-architecture equal those of the reference-class model for everything a
  released ``model_RGB2*.pth`` carries under ``backbone.*`` / ``sem_seg_head.*`` (checkpoint/odise_checkpointer.py:45-102
  loads with strict=False name matching);
* ``LdmRocm.exclude_unused_params`` freezes exactly the parameters the reference's dummy-backward procedure
  (ldm_diffusers.py:123-141) finds unused, for the shipped taps and for [5, 8] / 'in'-type taps.
"""
import os

import pytest
import torch

from oracle import sd_modules, ref_driver, madm_path
from madm_amd import weights


@pytest.fixture(scope="module")
def oracle_nets():
    vae = weights.synth_init_(sd_modules.AutoencoderKL(), 0, "vae.")
    unet = weights.synth_init_(sd_modules.UNet2DConditionModel(), 0, "unet.")
    return vae, unet


def test_load_diffusers_snapshot(tmp_path, oracle_nets):
    from safetensors.torch import save_file
    from madm_amd.ldm_rocm import LdmRocm
    vae, unet = oracle_nets
    os.makedirs(tmp_path / "unet")
    os.makedirs(tmp_path / "vae")
    save_file({k: v.contiguous() for k, v in unet.state_dict().items()}, str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors"))
    old = {}
    ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    n_renamed = 0
    for k, v in vae.state_dict().items():
        if ".attentions." in k:
            for new, o in ren.items():
                if f".{new}." in k:
                    k = k.replace(f".{new}.", f".{o}.")
                    if v.dim() == 2:
                        v = v[:, :, None, None]          # SD-v1-4's VAE stores these as 1x1 convs
                    n_renamed += 1
                    break
        old[k] = v.contiguous()
    assert n_renamed == 16                               # encoder + decoder mid attention: 4 layers x (weight, bias) x 2
    torch.save(old, str(tmp_path / "vae" / "diffusion_pytorch_model.bin"))   # the .bin fallback path
    unc = 0.02 * torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(5))
    torch.save(unc, str(tmp_path / "uncond_inputs.pt"))
    m = LdmRocm(str(tmp_path), [], [5, 8, 11], (), input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                weights='pretrained', device='cpu')
    for name, ref in (("unet", unet), ("vae", vae)):
        got = dict(getattr(m, name).named_parameters())
        want = dict(ref.named_parameters())
        assert set(got) == set(want)
        for k in want:
            assert torch.equal(got[k].detach(), want[k].detach()), (name, k)
    assert torch.equal(m.uncond_inputs, unc)
    # a key the snapshot does not have / an extra one is an error, not a silent skip
    bad = dict(unet.state_dict())
    bad.pop("conv_in.bias")
    save_file({k: v.contiguous() for k, v in bad.items()}, str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors"))
    with pytest.raises(RuntimeError, match="missing"):
        LdmRocm(str(tmp_path), [], [5, 8, 11], (), weights='pretrained', device='cpu')


def _product_model(device="cpu"):
    from madm_amd.ldm_rocm import LdmRocm
    from madm_amd.backbone import BasePromptTimeGenerator, AttentionFeatureExtractorBackbone
    from madm_amd.head import DAFormerHead
    from madm_amd.criterion import CmdiseCriterion
    from madm_amd.mtmadise import MTMADISE
    cfg = madm_path.DEPTH_CFG
    ldm = LdmRocm("", [], [5, 8, 11], (), input_range='-1+1', unet_block_indices_type='after', finetune_unet='all',
                  weights='synthetic', seed=0, vae_decoder_loss=True, device=device)
    gen = BasePromptTimeGenerator(ldm_extractor=ldm, same_cond_params=True)
    backbone = AttentionFeatureExtractorBackbone(None, list(cfg["feature_dims"]), None, feature_extractor=gen,
                                                 out_features=list(cfg["out_features"]),
                                                 projection_dim=list(cfg["projection_dim"]))
    head = DAFormerHead(in_channels=list(cfg["head_in_channels"]), in_keys=list(cfg["out_features"]), in_index=[0, 1, 2, 3],
                        channels=256, num_classes=11, decoder_params=madm_path.head_decoder_params())
    return MTMADISE(backbone, head, CmdiseCriterion(num_classes=11), target_modality="Depth", train_palette=[0] * 33,
                    vae_decoder_loss='st', vae_decoder_loss_type='L1')


@pytest.mark.skipif(not ref_driver.available(), reason="needs /root/reference (build container)")
def test_state_dict_keys_match_reference_classes(oracle_nets):
    """Keys (and shapes) under backbone.* / sem_seg_head.* / ema_sem_seg_head.* of the product model vs a model assembled
    from the REFERENCE's BasePromptTimeGenerator / AttentionFeatureExtractorBackbone / DAFormerHead classes on the
    diffusers-named oracle UNet / VAE -- what ODISECheckpointer matches by name when a released checkpoint is loaded."""
    from copy import deepcopy
    vae, unet = oracle_nets
    ns = ref_driver.load_modeling()
    backbone, head = madm_path.build_reference_eval_model(ns, vae, unet, madm_path.DEPTH_CFG)
    ref = torch.nn.Module()
    ref.backbone, ref.sem_seg_head = backbone, head
    ref.sem_seg_head_sec_modal = head                      # cmdise.py:153-156: the same object under a second name
    # CMDISE._inti_ema_weights (cmdise.py:307-335)
    ref.backbone.ema_feature_projections = deepcopy(backbone.feature_projections)
    ref.ema_sem_seg_head = deepcopy(head)
    ref.backbone.feature_extractor.ema_clip_project_others = deepcopy(backbone.feature_extractor.clip_project_others)
    want = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    got = {k: tuple(v.shape) for k, v in _product_model().state_dict().items()}
    skip = ("shared_noise", "uncond_inputs", "num_batches_tracked")
    for k, shp in want.items():
        if any(s in k for s in skip):
            continue
        assert k in got, f"missing key {k}"
        assert got[k] == shp, (k, got[k], shp)
    extra = [k for k in got if k not in want and not any(s in k for s in skip)]
    assert not extra, extra[:8]
    # the UNet / VAE sub-trees use diffusers 0.25 names (odise_checkpointer.py:45-102, README.md:96)
    assert "backbone.feature_extractor.ldm_extractor.unet.down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight" in got
    assert "backbone.feature_extractor.ldm_extractor.vae.encoder.mid_block.attentions.0.to_q.weight" in got


@pytest.mark.skipif(not ref_driver.available(), reason="needs /root/reference (build container)")
@pytest.mark.parametrize("taps,kind", [([5, 8, 11], "after"), ([5, 8], "after"), ([2, 7], "after"), ([5, 8, 11], "in"),
                                       ([3, 9], "in")])
def test_exclude_unused_params_equals_reference_procedure(oracle_nets, taps, kind):
    """ldm_diffusers.py:123-141 on the CPU oracle (reference's own diffusion_unet, float32 instead of .cuda().half(), an
    8 x 8 latent -- the set of parameters without gradient does not depend on the size) vs the structural rule."""
    import torch.nn.functional as F
    from madm_amd.ldm_rocm import LdmRocm
    _, unet = oracle_nets
    ref = ref_driver.load()
    for p in unet.parameters():
        p.requires_grad = True
        p.grad = None
    g = torch.Generator().manual_seed(1)
    _, feats = ref.diffusion_unet(unet=unet, sample=torch.rand((1, 4, 8, 8), generator=g), timestep=torch.zeros(1).long(),
                                  encoder_hidden_states=torch.rand((1, 77, 768), generator=g), res_time_embedding=None,
                                  unet_block_indices=taps, unet_block_indices_type=kind)
    F.mse_loss(input=feats[-1], target=torch.ones_like(feats[-1])).backward()
    frozen_ref = {n for n, p in unet.named_parameters() if p.grad is None}
    for p in unet.parameters():
        p.grad = None
    m = LdmRocm("", [], taps, (), input_range='-1+1', unet_block_indices_type=kind, finetune_unet='all', weights='synthetic',
                device='cpu')
    frozen = {n for n, p in m.unet.named_parameters() if not p.requires_grad}
    assert frozen == frozen_ref, (sorted(frozen - frozen_ref)[:5], sorted(frozen_ref - frozen)[:5])
    assert {"conv_out.weight", "conv_norm_out.weight"} <= frozen
