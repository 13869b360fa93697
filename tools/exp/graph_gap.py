#!/usr/bin/env python3
"""Where do the ~0.5 ms at the head of every graph replay go?  Captures [tiny kernel, forward] and replays it;
run under rocprofv3 --kernel-trace and look at the first kernels of the last replay (tools/last_replay.py ... silu)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from madm_amd import ops
from madm_amd.ldm_rocm import LdmRocm

dev = torch.device("cuda")
m = LdmRocm("", [], [5, 8, 11], [], input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
            compute_dtype=torch.bfloat16, weights='synthetic', seed=0)
call = (bench.make_inputs(2, 512, dev), "rgb")
m(*call); m.check_input_range = False
tiny = torch.zeros(64, device=dev, dtype=torch.bfloat16)
nfwd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    m(*call)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    ops.silu(tiny)
    for _ in range(nfwd):
        out = m(*call)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.replay()
e1.record(); torch.cuda.synchronize()
print(f"{nfwd} forwards per graph: {e0.elapsed_time(e1) / 10 / nfwd:.3f} ms per forward")
