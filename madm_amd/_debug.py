"""Debug harnesses (never on in production; both are switched by environment variables read at import).

``MADM_DEBUG_POISON_HBM=1`` -- the HBM counterpart of ``_lib._PoisonedLib`` (LDS): every device buffer the package obtains
WITHOUT initialising it is filled with a signalling pattern before the kernel that is supposed to write it runs --
``torch.empty`` / ``empty_like`` / ``empty_strided`` / ``Tensor.new_empty`` on a HIP device (so also every reuse of a block
from torch's caching allocator), and the per-stream split-K workspace on EVERY hand-out (``ops._workspace``).  Floating
types get quiet NaNs, byte workspaces 0xFF (a NaN in every float reading: f32 0xFFFFFFFF, f16 / bf16 0xFFFF), integer
types 0x7F7F... .  A kernel that reads a padded column, a partial tile's tail or a workspace slab nobody wrote then fails
its parity test deterministically instead of depending on what the previous owner of the block left there (the signature
of "first process on a fresh box", profiles/round5_f32_eval_transient.txt).  The fills are ordinary kernels on the current
stream, so they are captured into hipGraphs as nodes: a graph replay re-poisons its own intermediates every time.

``MADM_DEBUG_POISON_HBM=2`` additionally fills with a LARGE FINITE pattern instead of NaN (f32 1e30, f16 6e4, bf16 1e30):
NaN x 0 and 1e30 x 0 differ -- a zero-weighted read of a padded column survives the finite pattern and not the NaN one,
which tells "read but multiplied by zero" from "read and used".
"""
import os

MODE = int(os.environ.get("MADM_DEBUG_POISON_HBM", "0") or 0)
COUNT = {"tensors": 0, "bytes": 0, "workspace": 0}
_installed = False


def _fill(t):
    import torch
    if not t.is_cuda or t.numel() == 0:
        return t
    if t.dtype.is_floating_point:
        if MODE == 2:
            t.fill_(6.0e4 if t.dtype == torch.float16 else 1.0e30)
        else:
            t.fill_(float("nan"))
    elif t.dtype == torch.uint8:
        t.fill_(0xFF)
    elif t.dtype == torch.bool:
        return t
    elif t.dtype in (torch.int64, torch.int32, torch.int16, torch.int8):
        t.fill_({torch.int64: 0x7F7F7F7F7F7F7F7F, torch.int32: 0x7F7F7F7F, torch.int16: 0x7F7F, torch.int8: 0x7F}[t.dtype])
    else:
        return t
    COUNT["tensors"] += 1
    COUNT["bytes"] += t.numel() * t.element_size()
    return t


def poison(t):
    """Poisons one tensor when the harness is on (ops._workspace calls this on every hand-out)."""
    if MODE:
        COUNT["workspace"] += 1
        _fill(t)
    return t


def install():
    """Wraps torch's uninitialised-allocation entry points (idempotent).  Called by ``madm_amd/__init__`` when MODE != 0."""
    global _installed
    if _installed or not MODE:
        return
    import torch
    _installed = True

    def wrap(fn):
        def inner(*a, **k):
            return _fill(fn(*a, **k))
        inner.__name__ = getattr(fn, "__name__", "empty")
        inner.__wrapped__ = fn
        return inner

    torch.empty = wrap(torch.empty)
    torch.empty_like = wrap(torch.empty_like)
    torch.empty_strided = wrap(torch.empty_strided)
    torch.Tensor.new_empty = wrap(torch.Tensor.new_empty)
