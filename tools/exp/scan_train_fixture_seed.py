#!/usr/bin/env python3
"""Build container only (imports the reference's head / criterion through tests/golden/gen_golden.py).

Finds, for a train-fixture variant, an input seed whose TEACHER decisions are not within fp32 noise: the minimum top-2
probability gap of the teacher's soft-max over all pixels (an argmax tie flips a pseudo label) and the distance of the
nearest max-probability to the confidence threshold (a crossing moves pseudo_weight by 1 / pixels).  ADVICE r4 (medium): the
round-4 lora fixtures hold a pixel with gap 8.8e-7 and one 1.06e-6 from the threshold, and their gates had to be loosened.
For the chosen seed a threshold in the widest gap of the sorted max-probabilities near 0.25 is proposed.

usage: scan_train_fixture_seed.py <variant> [first_seed] [count]"""
import os
import sys
import random

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import gen_golden  # noqa: E402
from golden_util import TRAIN_CASE, train_inputs, train_dropout_scales  # noqa: E402

variant = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 8900
count = int(sys.argv[3]) if len(sys.argv) > 3 else 8
torch.set_num_threads(os.cpu_count())
model = gen_golden.build_train_oracle(reference=True, variant=variant)
state = {k: v.clone() for k, v in model.state_dict().items()}
for seed in range(first, first + count):
    model.load_state_dict(state)                      # BatchNorm running statistics move in a train-mode pass
    model.train_iter_index = 0
    sc = train_dropout_scales(TRAIN_CASE["B"])
    model.sem_seg_head.dropout.scales = [sc[0], sc[1]]
    model.ema_sem_seg_head.dropout.scales = [sc[2]]
    random.seed(TRAIN_CASE["py_seed"])
    np.random.seed(TRAIN_CASE["np_seed"])
    case = dict(TRAIN_CASE, input_seed=seed)
    with torch.no_grad():
        model.forward_train(train_inputs(**case))
    el = model.last_step["ema_logits"].detach()
    x = torch.nn.functional.interpolate(el, size=(case["size"], case["size"]), mode="bilinear", align_corners=False)
    sm = torch.softmax(x.double(), 1)
    top = sm.topk(2, dim=1).values
    gap = (top[:, 0] - top[:, 1]).min().item()
    p = top[:, 0].flatten().sort().values
    near = p[(p > 0.22) & (p < 0.28)]
    d = near[1:] - near[:-1]
    i = int(d.argmax())
    thr = float((near[i] + near[i + 1]) / 2)
    print(f"seed {seed}: min top-2 gap {gap:.3e}; |p - 0.25| min {float((p - 0.25).abs().min()):.3e}; widest free interval near "
          f"0.25: threshold {thr:.6f} +- {float(d[i]) / 2:.3e}", flush=True)
