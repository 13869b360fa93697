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


def test_kernel_suite_under_lds_poison(cuda):
    env = dict(os.environ, MADM_DEBUG_POISON_LDS="1")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
           os.path.join(HERE, "test_ops_gpu.py"), os.path.join(HERE, "test_labels_gpu.py"),
           os.path.join(HERE, "test_parity_gpu.py") + "::test_golden", "-k", "not full_t0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE))
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, f"kernels depend on LDS they did not write (or the harness failed):\n{tail}\n{r.stderr[-2000:]}"
    assert " passed" in tail
