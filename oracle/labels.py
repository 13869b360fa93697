"""ORACLE (test infrastructure only -- never imported by madm_amd): CPU restatement of the label / pseudo-label
pipeline of the self-training step, each function citing the reference lines it follows.  Pinned against the
reference's OWN code in this container by tests/test_oracle.py::test_label_oracle_equals_reference_functions
(functions extracted from /root/reference by AST, no third-party stand-ins needed for them) and against the committed
fixture tests/golden/labels.npz generated from that same reference code (tests/golden/gen_golden.py::main_labels).
"""
import ast
import os

import numpy as np
import torch

REF_MTMADISE = "/root/reference/modeling/meta_arch/mtmadise.py"
REF_DACS = "/root/reference/utils/dacs_transforms.py"


def convert_label_to_rgb(label, palette):
    """mtmadise.py:159-175.  label: int64 [B, 1, H, W]; palette: 768 ints (zero padded, :97-103).  PIL's 'P' -> 'RGB'
    conversion is a 256-entry table lookup of the uint8-cast label."""
    lab = label.cpu().numpy()
    pal = np.zeros(768, dtype=np.uint8)
    pal[:len(palette)] = np.asarray(palette, dtype=np.uint8)
    lut = pal.reshape(256, 3)
    out = []
    for i in range(lab.shape[0]):
        idx = lab[i, 0].astype(np.uint8)                       # :167 astype(np.uint8)
        rgb = torch.from_numpy(lut[idx].transpose(2, 0, 1).copy())   # :170-171
        out.append((rgb / 255 - 0.5) / 0.5)                    # :172 (f32: uint8 tensor / int -> float32)
    return torch.stack(out, 0), (label != 255).float()         # :163, :174


def pseudo_labels(ema_logits, size, pseudo_threshold):
    """mtmadise.py:339-349 (up to pseudo_weight before the optional crop)."""
    x = torch.nn.functional.interpolate(ema_logits, size=size, mode="bilinear", align_corners=False)
    sm = torch.softmax(x.detach(), dim=1)
    prob, label = torch.max(sm, dim=1)
    large = prob.ge(pseudo_threshold).long() == 1
    val = torch.sum(large).item() / np.size(np.array(label.cpu()))
    return prob, label, val * torch.ones(prob.shape)


def generate_class_mask(label, classes):
    """dacs_transforms.py:92-97."""
    label, classes = torch.broadcast_tensors(label, classes.unsqueeze(1).unsqueeze(2))
    return label.eq(classes).sum(0, keepdims=True)


def get_class_masks(labels, rng=np.random):
    """dacs_transforms.py:81-90 (note :84: the candidate classes are those of the whole batch)."""
    masks = []
    for label in labels:
        classes = torch.unique(labels)
        n = classes.shape[0]
        choice = rng.choice(n, int((n + n % 2) / 2), replace=False)
        masks.append(generate_class_mask(label, classes[torch.Tensor(choice).long()]).unsqueeze(0))
    return masks


def one_mix(mask, data=None, target=None):
    """dacs_transforms.py:100-111."""
    if mask is None:
        return data, target
    if data is not None:
        m, _ = torch.broadcast_tensors(mask[0], data[0])
        data = (m * data[0] + (1 - m) * data[1]).unsqueeze(0)
    if target is not None:
        m, _ = torch.broadcast_tensors(mask[0], target[0])
        target = (m * target[0] + (1 - m) * target[1]).unsqueeze(0)
    return data, target


# ---- the reference's own functions, extracted by AST (this container only) -------------------------------------
def reference_available():
    return os.path.exists(REF_MTMADISE) and os.path.exists(REF_DACS)


def _extract(path, names, cls=None, extra=None):
    tree = ast.parse(open(path).read())
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    picked = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(picked) == len(names), (path, names)
    for n in picked:
        n.decorator_list = []
    mod = ast.Module(body=picked, type_ignores=[])
    ns = {"torch": torch, "np": np}
    ns.update(extra or {})
    exec(compile(mod, path, "exec"), ns)
    return [ns[n] for n in names]


def reference_functions():
    """(convert_label_to_rgb, get_class_masks, generate_class_mask, one_mix) exactly as the reference defines them."""
    from PIL import Image
    (conv,) = _extract(REF_MTMADISE, ["convert_label_to_rgb"], cls="MTMADISE", extra={"Image": Image})
    gcm, gen, mix = _extract(REF_DACS, ["get_class_masks", "generate_class_mask", "one_mix"])
    return conv, gcm, gen, mix
