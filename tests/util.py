"""Test plumbing: layout helpers (torch on host) and error metrics."""
import torch


def to_tokens(x_nchw, dtype, cpad=None, device="cuda"):
    """NCHW f32 (CPU) -> channels-last [B*H*W, Cpad] device tensor."""
    B, C, H, W = x_nchw.shape
    t = x_nchw.permute(0, 2, 3, 1).reshape(B * H * W, C)
    if cpad is not None and cpad != C:
        t = torch.nn.functional.pad(t, (0, cpad - C))
    return t.to(dtype).contiguous().to(device)


def from_tokens(t, B, H, W):
    """[B*H*W, C] device tensor -> NCHW f32 CPU."""
    C = t.shape[1]
    return t.float().cpu().reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()


def rel_err(a, b):
    """max |a-b| / max |b| and relative L2."""
    a = a.double()
    b = b.double()
    d = (a - b).abs().max().item()
    s = b.abs().max().item() + 1e-30
    l2 = ((a - b).norm() / (b.norm() + 1e-30)).item()
    return d / s, l2


def bf16_round(x):
    return x.to(torch.bfloat16).float()
