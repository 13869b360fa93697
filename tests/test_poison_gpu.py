"""History independence of the kernels: a subset of the GPU suite re-run with every launch through the C ABI preceded by an
LDS poison (madm_amd/_lib.py::_PoisonedLib, MADM_DEBUG_POISON_LDS=1: all 160 KB of every CU hold quiet NaNs when a kernel
starts).  A kernel that reads LDS it has not written -- padding rows of a tile, a slot its DMA skipped, statistics scratch --
would pass or fail depending on what ran on the CU before it; here it fails deterministically.  (Round 5: one unexplained f32
mismatch of an eval forward, once, first process on a fresh box -- profiles/round5_f32_eval_transient.txt; the whole GPU suite
passes under the poison, so an unwritten-LDS read is not the cause.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_under(env_extra, targets, extra=()):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", *targets, *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE))
    return r, "\n".join(r.stdout.splitlines()[-25:])


def test_kernel_suite_under_lds_and_hbm_poison(cuda):
    """Both harnesses at once (they are independent; one subprocess instead of two keeps the suite's time down): every launch
    through the C ABI is preceded by an LDS poison (all 160 KB of every CU hold quiet NaNs when the kernel starts), and every
    uninitialised device allocation of the process + the split-K workspace on every hand-out hold NaNs until written
    (madm_amd/_debug.py, MADM_DEBUG_POISON_HBM=1; graph captures carry the fills as nodes).  A kernel that reads LDS it did not
    write, a padded column, a partial tile's tail or a workspace slab nobody wrote fails its parity test deterministically
    (VERDICT r5 'do this' 1; tools/exp/r6_poison_hbm.sh ran the WHOLE suite under the HBM poison + the fresh-process soak:
    profiles/round6_hbm_poison.txt)."""
    targets = [os.path.join(HERE, "test_ops_gpu.py"), os.path.join(HERE, "test_labels_gpu.py"),
               os.path.join(HERE, "test_parity_gpu.py") + "::test_golden"]
    r, tail = _run_under({"MADM_DEBUG_POISON_LDS": "1", "MADM_DEBUG_POISON_HBM": "1"}, targets, ["-k", "not full_t0"])
    assert r.returncode == 0, f"a kernel reads LDS / HBM nobody wrote (or a harness failed):\n{tail}\n{r.stderr[-2000:]}"
    assert " passed" in tail


def test_kernel_suite_under_hbm_poison_huge_finite(cuda):
    """MADM_DEBUG_POISON_HBM=2: 1e30 / 6e4 instead of NaN -- NaN x 0 and 1e30 x 0 differ: a zero-weighted read of a padded column
    survives the finite pattern and not the NaN one (which the test above covers), so the two modes together tell 'read but
    multiplied by zero' from 'read and used'.  The kernel-level file only (suite time)."""
    r, tail = _run_under({"MADM_DEBUG_POISON_HBM": "2"}, [os.path.join(HERE, "test_ops_gpu.py")])
    assert r.returncode == 0, f"a kernel reads HBM nobody wrote (or the harness failed):\n{tail}\n{r.stderr[-2000:]}"
    assert " passed" in tail


def test_hbm_poison_harness_is_effective(cuda):
    """Negative control of the harness itself: under MADM_DEBUG_POISON_HBM=1 an uninitialised device allocation holds NaNs, the
    split-K workspace is re-poisoned on every hand-out, a graph replay re-poisons its captured intermediates, and a kernel
    that reads a buffer nobody wrote propagates the NaN -- i.e. a green suite under the harness means something."""
    code = r'''
import torch, madm_amd
from madm_amd import _debug, ops
assert _debug.MODE == 1
x = torch.empty((64, 64), device="cuda", dtype=torch.float16)
assert torch.isnan(x).all()
assert torch.isnan(torch.empty_like(x)).all() and torch.isnan(x.new_empty((3,))).all()
assert int(torch.empty(4, dtype=torch.int32, device="cuda")[0]) == 0x7F7F7F7F
ws = ops._workspace(1 << 16, x.device); ws.zero_(); ws = ops._workspace(1 << 16, x.device)
assert int(ws[0]) == 0xFF and int(ws[-1]) == 0xFF
# a GEMM over an uninitialised operand: the NaNs must reach the output (zero weights do not hide them)
w = torch.zeros((64, 64), device="cuda", dtype=torch.float16)
out = ops.linear(torch.empty((128, 64), device="cuda", dtype=torch.float16), w)
assert torch.isnan(out).all()
# a captured graph re-poisons its intermediates at every replay
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    keep = torch.ones(8, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        t = torch.empty(8, device="cuda")
        keep.copy_(t)
    t.zero_(); g.replay(); torch.cuda.synchronize()
assert torch.isnan(keep).all()
n0 = _debug.COUNT["tensors"]
assert n0 >= 6
print("harness ok", _debug.COUNT)
'''
    env = dict(os.environ, MADM_DEBUG_POISON_HBM="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE))
    assert r.returncode == 0 and "harness ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
