"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/madm_hip.h
declares (no compute without a GPU), argument validation fails loudly with a message, weight packing,
parameter-name / count compatibility of the HIP modules with the oracle (diffusers naming), the
deterministic synthetic parameters, and that the product path has no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from madm_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "madm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(madm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.lib.madm_abi_version() == 5


def test_struct_layout_matches_header_field_order():
    from madm_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "madm_hip.h")).read()
    body = hdr[hdr.index("typedef struct {", hdr.index("madm_conv2d_fwd: implicit")):hdr.index("} madm_conv2d_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct {", "").strip()
        if not decl:
            continue
        names = [re.sub(r"[\*\s]", "", n.split()[-1]) for n in decl.split(",")]
        fields.extend(names)
    assert fields == [f[0] for f in _lib.Conv2dArgs._fields_], fields


def test_argument_validation_reports_errors_without_gpu():
    from madm_amd._lib import lib, Conv2dArgs, AttentionArgs
    a = Conv2dArgs()
    a.dtype = 1
    assert lib.madm_conv2d_fwd(ctypes.byref(a), None) == -1
    assert b"null tensor" in lib.madm_last_error()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf)
    a.in1 = a.w = a.out = p
    a.C1, a.B, a.IH, a.IW, a.OH, a.OW, a.KH, a.KW, a.stride, a.N, a.ldo, a.splitk = 48, 1, 1, 1, 1, 1, 1, 1, 1, 4, 4, 1
    assert lib.madm_conv2d_fwd(ctypes.byref(a), None) == -1
    assert b"multiple of 64" in lib.madm_last_error()
    a.C1, a.N = 64, 6
    assert lib.madm_conv2d_fwd(ctypes.byref(a), None) == -1 and b"multiple of 4" in lib.madm_last_error()
    a.N, a.splitk = 4, 2
    a.KH = a.KW = 3
    assert lib.madm_conv2d_fwd(ctypes.byref(a), None) == -1 and b"workspace" in lib.madm_last_error()
    at = AttentionArgs()
    at.q = at.k = at.v = at.o = p
    at.dtype, at.B, at.H, at.Lq, at.Lk, at.D = 1, 1, 1, 4, 4, 48
    at.ldq = at.ldk = at.ldv = at.ldo = 48
    assert lib.madm_attention_fwd(ctypes.byref(at), None) == -2 and b"not instantiated" in lib.madm_last_error()
    assert lib.madm_layernorm_fwd(1, None, None, 1, 8, None, None, 1e-5, None) == -1
    assert lib.madm_groupnorm_stats(0, p, 1, 1, 6, p, None) == -1     # C not a multiple of 4


def _header_struct_fields(struct_name):
    hdr = open(os.path.join(ROOT, "include", "madm_hip.h")).read()
    end = hdr.index("} " + struct_name + ";")
    body = hdr[hdr.rindex("typedef struct {", 0, end):end]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct {", "").strip()
        if decl:
            fields.extend(re.sub(r"[\*\s]", "", n.split()[-1]) for n in decl.split(","))
    return fields


def test_gradient_struct_layouts_match_header():
    from madm_amd import _lib
    assert _header_struct_fields("madm_conv2d_wgrad_args") == [f[0] for f in _lib.Conv2dWgradArgs._fields_]
    assert _header_struct_fields("madm_attention_bwd_args") == [f[0] for f in _lib.AttentionBwdArgs._fields_]
    assert _header_struct_fields("madm_attention_args") == [f[0] for f in _lib.AttentionArgs._fields_]


def test_gradient_entry_points_validate_arguments_without_gpu():
    from madm_amd._lib import lib, Conv2dWgradArgs, AttentionBwdArgs
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf)
    a = Conv2dWgradArgs()
    assert lib.madm_conv2d_wgrad(ctypes.byref(a), None) == -1 and b"null argument" in lib.madm_last_error()
    a.in1 = a.dout = a.dw = p
    a.dtype, a.C1, a.N = 1, 12, 8
    a.B = a.IH = a.IW = a.OH = a.OW = a.KH = a.KW = a.stride = 1
    assert lib.madm_conv2d_wgrad(ctypes.byref(a), None) == -1 and b"multiples of 8" in lib.madm_last_error()
    a.C1, a.N = 16, 6
    assert lib.madm_conv2d_wgrad(ctypes.byref(a), None) == -1 and b"multiple of 4" in lib.madm_last_error()
    b = AttentionBwdArgs()
    for f in ("q", "k", "v", "o", "dout", "dq", "dk", "dv"):
        setattr(b, f, p)
    b.dtype, b.B, b.H, b.Lq, b.Lk, b.D = 1, 1, 2, 4, 4, 40
    b.ldq = b.ldk = b.ldv = b.ldo = b.lddo = b.lddq = b.lddk = b.lddv = 80
    assert lib.madm_attention_bwd_workspace_bytes(ctypes.byref(b)) == 2 * 1 * 2 * 4 * 4
    assert lib.madm_attention_bwd(ctypes.byref(b), None) == -1 and b"workspace" in lib.madm_last_error()
    b.workspace, b.workspace_bytes, b.D = p, 64, 48
    b.ldq = b.ldk = b.ldv = b.ldo = b.lddo = b.lddq = b.lddk = b.lddv = 96
    assert lib.madm_attention_bwd(ctypes.byref(b), None) == -2 and b"not instantiated" in lib.madm_last_error()
    assert lib.madm_layernorm_bwd(1, p, p, p, 4, 12, p, 1e-5, None, None, None, None) == -1   # C not a multiple of 8
    assert lib.madm_geglu_bwd(1, p, p, p, 1, 12, None) == -1
    assert lib.madm_colsum(1, p, 8, 1, 1, 12, p, None) == -1


def test_weight_gradient_unpacking_inverts_the_weight_packing():
    from madm_amd import packing
    w = torch.randn(8, 96, 3, 3)
    for splits, kt in ((None, 32), ([64, 32], 32), ([40, 56], 64)):
        p = packing.pack_conv_weight(w, torch.float32, kt, splits=splits)
        back = packing.unpack_conv_weight_grad(p, 96, 3, 3, kt, splits=splits)
        assert torch.equal(back, w), (splits, kt)
    w5 = torch.randn(4, 5, 3, 3)   # tiny channel count: the padding channels are dropped
    assert torch.equal(packing.unpack_conv_weight_grad(packing.pack_conv_weight(w5, torch.float32, 32), 5, 3, 3, 32), w5)


def test_ops_refuse_cpu_tensors():
    from madm_amd import ops
    x = torch.zeros(4, 64)
    with pytest.raises(ValueError, match="no CPU fallback"):
        ops.layernorm(x, torch.ones(64), torch.zeros(64), 1e-5)
    with pytest.raises(ValueError, match="no CPU fallback"):
        ops.conv2d(x, torch.zeros(4, 64), 1, 4, 1, N=4)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under madm_amd/ may import or execute it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "madm_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "import oracle" not in src and "from oracle" not in src, f


def test_packing_layouts():
    from madm_amd import packing
    w = torch.arange(2 * 5 * 3 * 3, dtype=torch.float32).reshape(2, 5, 3, 3)
    p = packing.pack_conv_weight(w, torch.float32, 32)
    assert p.shape == (2, 9 * 32)
    p4 = p.reshape(2, 3, 3, 32)
    assert torch.equal(p4[..., :5], w.permute(0, 2, 3, 1)) and p4[..., 5:].abs().sum() == 0
    w2 = torch.randn(4, 96, 3, 3)
    p2 = packing.pack_conv_weight(w2, torch.float32, 32, splits=[64, 32]).reshape(4, 3, 3, 96)
    assert torch.equal(p2, w2.permute(0, 2, 3, 1))
    wl, bl = torch.randn(8, 6), torch.randn(8)
    pg, bg = packing.pack_geglu_weight(wl, bl, torch.float32, 32)
    assert torch.equal(pg[0::2, :6], wl[:4]) and torch.equal(pg[1::2, :6], wl[4:])
    assert torch.equal(bg[0::2], bl[:4]) and torch.equal(bg[1::2], bl[4:])
    assert packing.pack_linear_weight(torch.randn(3, 70), torch.bfloat16, 64).shape == (3, 128)
    # folded LayerNorm: Linear(LN(x)) == rstd (x W'^T - mean colsum) + bias'; ``interleave`` == the GEGLU row order first
    g = torch.Generator().manual_seed(3)
    x, w, b = torch.randn((5, 64), generator=g), torch.randn((8, 64), generator=g), torch.randn((8,), generator=g)
    gamma, beta = 1 + 0.1 * torch.randn((64,), generator=g), 0.1 * torch.randn((64,), generator=g)
    W, B, C = packing.fold_layernorm(w, b, gamma, beta, torch.float32, 32)
    mean, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    got = (x @ W.t() - mean * C[None, :]) / torch.sqrt(var + 1e-5) + B[None, :]
    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (64,), gamma, beta, 1e-5), w, b)
    assert torch.allclose(got, ref, atol=2e-5)
    Wi, Bi, Ci = packing.fold_layernorm(w, b, gamma, beta, torch.float32, 32, interleave=True)
    assert torch.equal(Wi[0::2], W[:4]) and torch.equal(Wi[1::2], W[4:]) and torch.equal(Bi[0::2], B[:4])
    assert torch.equal(Ci[1::2], C[4:])


def test_pack_entry_points_validate_arguments_without_gpu():
    from madm_amd._lib import lib
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf)
    one = (ctypes.c_int * 1)(8)
    assert lib.madm_pack_weight(1, p, p, 64, 4, 8, 25, 1, one, 64, 0, None) == -1 and b"taps <= 9" in lib.madm_last_error()
    assert lib.madm_pack_weight(1, p, p, 64, 4, 9, 1, 1, one, 64, 0, None) == -1 and b"sources hold" in lib.madm_last_error()
    assert lib.madm_pack_weight(1, p, p, 32, 4, 8, 1, 1, one, 64, 0, None) == -1 and b"ldo" in lib.madm_last_error()
    assert lib.madm_pack_weight(1, p, p, 64, 3, 8, 1, 1, one, 64, 1, None) == -1 and b"even row count" in lib.madm_last_error()
    assert lib.madm_fold_layernorm_pack(1, p, None, None, p, p, p, p, 4, 64, 0, None) == -1
    a = __import__("madm_amd._lib", fromlist=["Conv2dArgs"]).Conv2dArgs()
    assert lib.madm_conv2d_can_post_groupnorm(ctypes.byref(a)) == 0          # no split-K, no groups: cannot carry the norm
    a.dtype, a.splitk, a.pn_groups, a.N, a.C1, a.KH, a.KW, a.OH, a.OW, a.IH, a.IW, a.B = 1, 4, 32, 1280, 512, 1, 1, 16, 16, 16, 16, 2
    assert lib.madm_conv2d_can_post_groupnorm(ctypes.byref(a)) == 1          # 16 x 16 x 40 channels x 4 B = 40 KB of LDS
    a.OH = a.OW = a.IH = a.IW = 64
    assert lib.madm_conv2d_can_post_groupnorm(ctypes.byref(a)) == 0          # 640 KB: the group does not fit
    assert lib.madm_groupnorm_apply_cat(1, p, p, p, 64, 1, 1, 32, 32, 32, None, p, p, p, 1e-5, 0, None) == -1


@pytest.fixture(scope="module")
def trees():
    from oracle import sd_modules
    from madm_amd import sd_unet, sd_vae
    return (sd_unet.UNet2DConditionModel(), sd_vae.AutoencoderKL(), sd_modules.UNet2DConditionModel(),
            sd_modules.AutoencoderKL())


def test_parameter_names_and_counts_match_diffusers_layout(trees):
    unet, vae, o_unet, o_vae = trees
    assert sum(p.numel() for p in unet.parameters()) == 859_520_964
    assert sum(p.numel() for p in vae.parameters()) == 83_653_863
    for a, b in ((unet, o_unet), (vae, o_vae)):
        sa = {k: tuple(v.shape) for k, v in a.state_dict().items()}
        sb = {k: tuple(v.shape) for k, v in b.state_dict().items()}
        assert sa == sb
    keys = set(unet.state_dict())
    for k in ("down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight",
              "up_blocks.3.attentions.2.transformer_blocks.0.ff.net.0.proj.bias",
              "mid_block.resnets.1.time_emb_proj.weight", "time_embedding.linear_2.bias",
              "up_blocks.0.upsamplers.0.conv.weight", "conv_norm_out.weight"):
        assert k in keys, k
    vkeys = set(vae.state_dict())
    for k in ("encoder.mid_block.attentions.0.to_q.bias", "encoder.down_blocks.2.downsamplers.0.conv.weight",
              "quant_conv.weight", "decoder.up_blocks.3.resnets.2.conv2.bias", "post_quant_conv.bias"):
        assert k in vkeys, k


def test_lora_wrapping_matches_peft_shape(trees):
    import copy
    from types import SimpleNamespace
    unet = copy.deepcopy(trees[0])
    o_unet = copy.deepcopy(trees[2])
    from oracle import sd_modules
    cfg = SimpleNamespace(r=4, lora_alpha=4, target_modules=["to_k", "to_q", "to_v", "to_out.0"])
    base = sum(p.numel() for p in unet.parameters())
    unet.add_adapter(cfg, "Depth")
    o_unet.add_adapter(sd_modules.LoraConfig(r=4, lora_alpha=4), "Depth")
    assert sum(p.numel() for p in unet.parameters()) - base == 199_296 * 4
    assert set(unet.state_dict()) == set(o_unet.state_dict())
    names = [n for n, _ in unet.named_parameters() if "lora" in n]
    assert len(names) == 256 and all("Depth" in n for n in names)
    # the reference flips adapters by attribute (mtmadise.py:144-147)
    mods = [m for m in unet.modules() if hasattr(m, "_active_adapter")]
    assert len(mods) == 128
    unet.set_adapter(["Depth"])
    assert all(m._active_adapter == ["Depth"] for m in mods)


def test_synthetic_parameters_are_name_keyed_and_deterministic(trees):
    from madm_amd import weights
    _, vae, _, o_vae = trees
    weights.synth_init_(vae, 0, "vae.")
    weights.synth_init_(o_vae, 0, "vae.")
    sa, sb = vae.state_dict(), o_vae.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    w0 = sa["encoder.conv_in.weight"].clone()
    weights.synth_init_(vae, 1, "vae.")
    assert not torch.equal(vae.state_dict()["encoder.conv_in.weight"], w0)
    assert abs(w0.std().item() - (27 ** -0.5)) < 0.02


def test_ldm_rocm_keeps_ldm_diffusers_class_surface():
    """Static attributes BasePromptTimeGenerator / FeatureExtractorBackbone read (ldm_base.py:769-774,940-960;
    feature_extractor.py:89-113) -- compared with the reference class body when /root/reference exists."""
    from madm_amd.ldm_rocm import LdmRocm
    assert LdmRocm.text_embed_shape == torch.Size([77, 768]) and LdmRocm.unet_time_embed_out_features == 1280
    assert LdmRocm.uncond_inputs_size == torch.Size([1, 77, 768]) and LdmRocm.latent_image_size == (64, 64)
    ref = "/root/reference/modeling/meta_arch/ldm_diffusers.py"
    if os.path.exists(ref):
        src = open(ref).read()
        for attr in ("feature_size", "feature_dims", "feature_strides", "num_groups", "grouped_indices", "timesteps",
                     "input_mean", "input_std"):
            m = re.search(rf"^\s+{attr}\s*=\s*(.+)$", src, flags=re.M)
            assert m and eval(m.group(1)) == getattr(LdmRocm, attr), attr
        import inspect
        sig = list(inspect.signature(LdmRocm.__init__).parameters)
        m = re.search(r"def __init__\(self, (.*?)\):", src, flags=re.S)
        ref_args = [a.split("=")[0].strip() for a in m.group(1).replace("\n", " ").split(",")]
        assert sig[1:1 + len(ref_args)] == ref_args
    with pytest.raises(NotImplementedError):
        LdmRocm("", [], [5, 8, 11], [], concat_pixel_shuffle=True, weights="synthetic", device="cpu")


def test_bench_gpus_n_spawns_the_ranks_before_touching_a_gpu():
    """``python bench.py --gpus 2`` outside a torchrun environment starts 2 child ranks through torch.distributed.run with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (--dry-launch: the ranks report and exit before any GPU call); without
    enough visible devices the launcher refuses with a clear message instead of running one rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    ranks = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")), key=lambda d: d["RANK"])
    assert [d["RANK"] for d in ranks] == ["0", "1"] and [d["LOCAL_RANK"] for d in ranks] == ["0", "1"]
    assert all(d["WORLD_SIZE"] == "2" and d["MASTER_ADDR"] == "127.0.0.1" and d["MASTER_PORT"] for d in ranks)
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                           timeout=300, env=env)
        assert r.returncode != 0 and "2 GPUs requested" in r.stderr and "visible" in r.stderr


def test_shipped_code_has_no_low_lane_op_sel_packed_fp32():
    """on MI355X `v_pk_add_f32 ... op_sel:[0,1]` (LOW result fed from the HIGH register of a pair)
    intermittently read that operand as 0 in lanes 48..63 beside a co-resident MFMA wave.  The built library must not contain
    that operand form at all, and packed-FP32 ops only in the 16 x 16 halo conv (csrc/Makefile builds everything else with
    -packed-fp32-ops).  Checked on the binary: gfx950 code objects cut out of libmadm_hip.so and disassembled."""
    import importlib.util
    import shutil
    from madm_amd._lib import LIB_PATH
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("isa_pk_scan", os.path.join(root, "tools", "isa_pk_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not (os.path.exists(mod.OBJDUMP) or shutil.which(mod.OBJDUMP)):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    total, low, per = mod.scan(LIB_PATH)
    assert len(mod.code_objects(LIB_PATH)) >= 10, "no gfx950 code objects found in the library"
    assert low == 0, f"{low} packed-FP32 instructions feed a LOW result from a HIGH register"
    # (mangled names: conv3x3_h16_kernelIDF16_... = f16, ...IDF16b... = bf16, ...kernelIf... = float: the f32 instantiation is
    # built without packed ops since round 6 -- conv3x3_h16_f32.hip)
    stray = {k: v for k, v in per.items() if "conv3x3_h16_kernelIDF16" not in k}
    assert not stray, f"packed-FP32 ops outside the 16-bit 16 x 16 halo conv: {list(stray.items())[:5]}"


def test_shipped_kernels_use_no_scratch():
    """Round 6: no kernel of the library may spill to scratch.  The generic epilogue of the 16 x 16 halo conv -- the one every
    f32-mode launch takes -- spilled 400 .. 750 VGPRs (private segment 680 .. 876 B per lane) until it was split into two-row
    steps; the only other scratch user the path ever had (the rejected weights-direct variant) was not bit-stable under the
    multi-stream pipeline.  Read from the code objects' metadata (tools/scratch_scan.py)."""
    import importlib.util
    import shutil
    from madm_amd._lib import LIB_PATH
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("scratch_scan", os.path.join(root, "tools", "scratch_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not (os.path.exists(mod.READELF) or shutil.which(mod.READELF)):
        pytest.skip("llvm-readelf of the ROCm toolchain not found")
    rows = mod.kernels(LIB_PATH)
    assert len(rows) >= 250, f"only {len(rows)} kernels found in the library's code objects"
    bad = {k: v for k, v in rows.items() if v.get("private_segment_fixed_size", 0) > 0 or v.get("vgpr_spill_count", 0) > 0}
    assert not bad, f"kernels with scratch: {[(k[:80], v.get('private_segment_fixed_size')) for k, v in bad.items()][:6]}"


def test_lora_targets_outside_the_attention_projections_are_refused():
    """the adapters run as K-extensions of the fused attention projections; a LoRA config whose target_modules
    match any other Linear (e.g. 'net.2', which Transformer2DModel composes with proj_out) must fail loudly at add_adapter,
    not with an AttributeError somewhere in the forward."""
    from types import SimpleNamespace
    import pytest
    from madm_amd import sd_unet
    t = sd_unet.Transformer2DModel(heads=2, dim_head=40, in_channels=80, cross_attention_dim=64)
    cfg = SimpleNamespace(r=4, lora_alpha=4, target_modules=["to_q", "net.2"])
    with pytest.raises(NotImplementedError, match="net.2"):
        sd_unet.UNet2DConditionModel.add_adapter(t, cfg, "x")
    t = sd_unet.Transformer2DModel(heads=2, dim_head=40, in_channels=80, cross_attention_dim=64)
    sd_unet.UNet2DConditionModel.add_adapter(t, SimpleNamespace(r=4, lora_alpha=4, target_modules=["to_k", "to_q", "to_v", "to_out.0"]), "x")
    wrapped = [n for n, m in t.named_modules() if isinstance(m, sd_unet.LoraLinear)]
    assert len(wrapped) == 8 and type(t.transformer_blocks[0].ff.net[2]) is sd_unet.Linear


def test_tuned_table_rows_are_valid_and_unique():
    """madm_amd/csrc/igemm_tuned.inc (tile / split-K rows chosen by tools/tune_concurrent.py): every row names an existing tile code
    and a split-K >= 1, no (dtype, M, N, K, KH, variant) key appears twice (the first match would silently win), and the rows of
    tile 12 / 9 / 10 (the halo kernels) are 3 x 3 shapes."""
    import collections
    import os
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "madm_amd", "csrc")

    def read(name):
        out = []
        for ln in open(os.path.join(csrc, name)):
            m = re.match(r"^\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\}", ln)
            if m:
                out.append(tuple(int(v) for v in m.groups()))
        dup = [k for k, n in collections.Counter(r[:6] for r in out).items() if n > 1]
        assert not dup, (name, dup[:5])
        return out

    rows = read("igemm_tuned.inc")
    assert len(rows) > 250
    lat = read("igemm_tuned_latency.inc")     # round 6: the latency profile's rows (madm_set_tuning_profile(1)), same format
    base = {r[:6]: r[6:] for r in rows}
    for r in lat:   # a latency row that repeats the throughput row is noise (and hides a stale pair when one of them is re-tuned)
        assert base.get(r[:6]) != r[6:], r
    for dtype, M, N, K, KH, variant, tile, sk in rows + lat:
        assert dtype in (0, 1) and M > 0 and N > 0 and K > 0 and KH in (1, 3) and 0 <= variant <= 3, (M, N, K)
        assert 1 <= tile <= 17 and sk >= 1, (M, N, K, tile, sk)
        if tile in (4, 5, 9, 10, 12):
            assert KH == 3, (M, N, K, tile)
        if variant in (1, 2, 3):
            assert KH == 3, (M, N, K, variant)


def test_tuning_profiles_switch_rows_and_pin():
    """Round 6: madm_set_tuning_profile(1) puts the lone-launch rows of igemm_tuned_latency.inc in front of the throughput
    table; ops.tuning_profile restores the previous profile on exit, and a pinned context (the graph runners: "throughput") makes the
    sync_profile() of a synchronous forward inside it a no-op.  Host-side only: pick_tile / suggest_splitk need no GPU."""
    import ctypes
    import re
    from madm_amd import ops
    from madm_amd._lib import lib, Conv2dArgs, MADM_F16
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "madm_amd", "csrc")

    def rows(name):
        out = {}
        for ln in open(os.path.join(csrc, name)):
            m = re.match(r"^\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\}", ln)
            if m:
                v = tuple(int(x) for x in m.groups())
                out[v[:6]] = v[6:]
        return out

    base, lat = rows("igemm_tuned.inc"), rows("igemm_tuned_latency.inc")
    assert len(lat) >= 20
    # a plain linear-layer shape (KH = 1, variant 0) whose two rows name different igemm tiles
    key = next(k for k in sorted(lat) if k[4] == 1 and k[5] == 0 and k in base and base[k][0] != lat[k][0]
               and lat[k][0] != 13 and base[k][0] != 13)
    _, M, N, K, _, _ = key
    a = Conv2dArgs()
    a.dtype, a.C1, a.C2, a.B, a.IH, a.IW, a.OH, a.OW = MADM_F16, K, 0, 1, M, 1, M, 1
    a.KH = a.KW = a.stride = 1
    a.N, a.splitk = N, 1
    a.in1 = a.w = a.out = 1
    pick = lambda: (lib.madm_conv2d_pick_tile(ctypes.byref(a)), lib.madm_conv2d_suggest_splitk(ctypes.byref(a)))
    assert lib.madm_get_tuning_profile() == 0 and pick() == base[key]
    with ops.tuning_profile("latency"):
        assert lib.madm_get_tuning_profile() == 1 and pick() == lat[key]
        with ops.tuning_profile("throughput", pin=True):           # a runner's capture
            assert pick() == base[key]
            with ops.sync_profile():                                # a forward() inside it: ignored
                assert lib.madm_get_tuning_profile() == 0 and pick() == base[key]
            assert lib.madm_get_tuning_profile() == 0
        assert lib.madm_get_tuning_profile() == 1
    assert lib.madm_get_tuning_profile() == 0 and ops.tuning_profile._pinned == 0
    with ops.sync_profile():
        assert lib.madm_get_tuning_profile() == (1 if ops.SYNC_PROFILE == "latency" else 0)
    assert lib.madm_set_tuning_profile(7) != 0 and b"profile" in lib.madm_last_error()
    assert lib.madm_get_tuning_profile() == 0


def test_stream_safe_cache_and_side_build_hooks_are_inert_on_the_host():
    """Round 6: the lazily-built-constant cache and the side-stream build hook must cost nothing where no second
    stream exists: CPU tensors carry no event, a hit returns the SAME object, ``note_build`` outside ``side_builds`` does nothing."""
    import torch
    from madm_amd import ops
    from madm_amd.ldm_rocm import _StreamSafeCache
    c = _StreamSafeCache()
    n0 = _StreamSafeCache.fills
    built = []

    def build():
        built.append(1)
        return torch.arange(4)

    a = c.get_or_build(("k", 1), build)
    b = c.get_or_build(("k", 1), build)
    assert a is b and len(built) == 1 and _StreamSafeCache.fills == n0 + 1 and c[("k", 1)][1] is None
    t = c.get_or_build(("pair",), lambda: (torch.ones(1), torch.zeros(1)))
    assert isinstance(t, tuple) and c.get_or_build(("pair",), lambda: None) is t
    before = ops.SIDE_BUILDS_NOTED
    ops.note_build()                                    # no side_builds context: nothing happens, no CUDA call
    assert ops.SIDE_BUILDS_NOTED == before and ops._SIDE_BUILD_MAIN is None
