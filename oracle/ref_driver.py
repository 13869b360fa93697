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


def load_modeling():
    """Loads the reference's OWN ``BasePromptTimeGenerator`` / ``ClipFeatureProject`` (modeling/meta_arch/ldm_base.py),
    ``AttentionFeatureExtractorBackbone`` (modeling/backbone/feature_extractor.py) and ``DAFormerHead``
    (modeling/sem_seg_head/daformer_head.py) from /root/reference, on top of the restated third-party classes of
    oracle/third_party.py and inert stubs for packages those files import but the hot path never calls.
    Returns a namespace with the three modules."""
    from . import third_party as tp
    root = "/root/reference"
    saved = {}

    def put(name, mod):
        saved[name] = sys.modules.get(name)
        sys.modules[name] = mod

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        put(name, m)
        return m

    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        put(name, m)
        return m

    try:
        # real reference packages, WITHOUT running their __init__ (which imports the whole training stack)
        pkg("modeling", root + "/modeling")
        pkg("modeling.meta_arch", root + "/modeling/meta_arch")
        pkg("modeling.backbone", root + "/modeling/backbone")
        pkg("modeling.sem_seg_head", root + "/modeling/sem_seg_head")
        # third-party packages: restatements (third_party.py) or inert names
        stub("torchvision")
        stub("torchvision.transforms", Resize=tp.Resize, InterpolationMode=tp.InterpolationMode)
        stub("detectron2")
        stub("detectron2.modeling")
        stub("detectron2.modeling.backbone", Backbone=tp.Backbone)
        stub("detectron2.modeling.backbone.resnet", BottleneckBlock=tp.BottleneckBlock, ResNet=tp.ResNet)
        stub("detectron2.structures", ImageList=tp.ImageList)
        stub("mmcv")
        stub("mmcv.cnn", ConvModule=tp.ConvModule, DepthwiseSeparableConvModule=tp.DepthwiseSeparableConvModule)
        stub("mmcv.runner", BaseModule=tp.BaseModule)
        oc = stub("omegaconf", OmegaConf=tp.OmegaConf)
        oc.listconfig = stub("omegaconf.listconfig", ListConfig=tp.ListConfig)
        stub("timm")
        stub("timm.models")
        stub("timm.models.layers", trunc_normal_=tp.trunc_normal_)
        for name, attrs in (("ldm", {}), ("ldm.models", {}), ("ldm.models.diffusion", {}),
                            ("ldm.models.diffusion.ddpm", {"LatentDiffusion": object}),
                            ("ldm.modules", {}), ("ldm.modules.diffusionmodules", {}),
                            ("ldm.modules.diffusionmodules.openaimodel", {"timestep_embedding": None}),
                            ("ldm.modules.distributions", {}),
                            ("ldm.modules.distributions.distributions", {"DiagonalGaussianDistribution": object}),
                            ("ldm.util", {"instantiate_from_config": None}),
                            ("checkpoint", {}), ("checkpoint.odise_checkpointer", {"LdmCheckpointer": object}),
                            ("modeling.meta_arch.clip", {"ClipAdapter": object}),
                            ("utils", {}), ("utils.file_io", {"PathManager": object})):
            stub(name, **attrs)
        import importlib
        ns = types.SimpleNamespace()
        ns.ldm_base = importlib.import_module("modeling.meta_arch.ldm_base")
        ns.feature_extractor = importlib.import_module("modeling.backbone.feature_extractor")
        ns.daformer_head = importlib.import_module("modeling.sem_seg_head.daformer_head")
        return ns
    finally:
        for k, old in saved.items():
            if old is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = old
        for k in [k for k in sys.modules if k.startswith("modeling.") and k not in saved]:
            sys.modules.pop(k, None)
