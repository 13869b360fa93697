"""Where the ATen glue launches of one training step come from: CPU-side torch.profiler (no device tracing) with Python
stacks, aggregated by (aten op, innermost madm_amd / bench frame).  usage: python tools/exp/train_aten_sites.py"""
import collections
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from madm_amd.train import MadmTrainer
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda")
model = bench.build_train_model(torch.float16, dev, False)
trainer = MadmTrainer(model, lr=5e-6, weight_decay=0.05, grad_clip=0.01, dist=None, amp=True)
data = bench.train_inputs(2, 512, dev)
for _ in range(2):
    trainer.run_step(data)
torch.cuda.synchronize()
# (torch 2.10: Python stacks of CPU events need the experimental verbose config)
try:
    _cfg = torch._C._profiler._ExperimentalConfig(verbose=True)
except Exception:
    _cfg = None
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True, experimental_config=_cfg) as prof:
    trainer.run_step(data)
    torch.cuda.synchronize()
LAUNCHING = {"aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::mul", "aten::add", "aten::add_", "aten::mul_",
             "aten::sub", "aten::div", "aten::div_", "aten::where", "aten::clamp", "aten::sum", "aten::mean", "aten::index",
             "aten::index_put_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::zeros", "aten::ones",
             "aten::full", "aten::eq", "aten::ne", "aten::lt", "aten::gt", "aten::ge", "aten::le", "aten::exp", "aten::sqrt",
             "aten::rsqrt", "aten::neg", "aten::abs", "aten::masked_fill_", "aten::bitwise_and", "aten::logical_and",
             "aten::_foreach_mul_", "aten::_foreach_add_", "aten::_foreach_norm", "aten::stack", "aten::repeat",
             "aten::expand", "aten::sigmoid", "aten::tanh", "aten::softmax", "aten::_softmax", "aten::argmax", "aten::max",
             "aten::min", "aten::pow", "aten::lerp_", "aten::addcmul_", "aten::addcdiv_", "aten::randn", "aten::rand",
             "aten::normal_", "aten::uniform_", "aten::bernoulli_", "aten::randint", "aten::linalg_vector_norm"}
agg = collections.Counter()
for ev in prof.events():
    if ev.name not in LAUNCHING:
        continue
    site = "?"
    for fr in ev.stack:
        if ("madm_amd/" in fr or "bench.py" in fr) and "site-packages" not in fr and "dist-packages" not in fr:
            site = fr[fr.index("madm_amd/"):] if "madm_amd/" in fr else fr[fr.index("bench.py"):]
            site = site[:90]
            break
    shp = str(ev.input_shapes)[:60] if ev.input_shapes else ""
    agg[(ev.name, site)] += 1
print(f"{sum(agg.values())} launching aten ops")
for (name, site), n in agg.most_common(70):
    print(f"{n:5d}  {name:22s} {site}")
