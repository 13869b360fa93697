"""ORACLE helper (this container only): loads the REFERENCE's own orchestration functions from
/root/reference/modeling/meta_arch/ldm_diffusers.py by file path, with stub ``sys.modules`` entries for
the third-party packages that are not installed.  The functions are duck-typed over
``unet.*`` / ``vae.*`` attributes, so they drive the oracle modules of oracle/sd_modules.py and thereby
pin tap indexing, skip popping, ``emb += cond_emb``, latent scaling and noise mixing to the reference's
code.  Never used on the GPU box (``/root/reference`` does not exist there) and never by the product.
"""
import importlib.util
import os
import sys
import types

REF_FILE = "/root/reference/modeling/meta_arch/ldm_diffusers.py"


def available():
    return os.path.exists(REF_FILE)


def load():
    from . import sd_modules
    saved = {}
    stubs = {}

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        stubs[name] = m
        return m

    class UNet2DConditionOutput:
        def __init__(self, sample):
            self.sample = sample

    stub("diffusers", AutoencoderKL=object, DDPMScheduler=object, UNet2DConditionModel=object)
    stub("diffusers.models")
    stub("diffusers.models.autoencoders")
    stub("diffusers.models.autoencoders.vae", DiagonalGaussianDistribution=sd_modules.DiagonalGaussianDistribution)
    stub("diffusers.models.unet_2d_condition", UNet2DConditionOutput=UNet2DConditionOutput)
    stub("modeling")
    stub("modeling.neti", NeTICLIPTextModel=object)
    need_tf = False
    try:
        from transformers import CLIPTokenizer  # noqa: F401
    except Exception:
        need_tf = True
    if need_tf:
        stub("transformers", CLIPTokenizer=object)
    for k, m in stubs.items():
        saved[k] = sys.modules.get(k)
        sys.modules[k] = m
    try:
        spec = importlib.util.spec_from_file_location("_madm_ref_ldm_diffusers", REF_FILE)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, old in saved.items():
            if old is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = old
    return mod
