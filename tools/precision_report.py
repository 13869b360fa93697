"""Per-tensor error of the 16-bit compute modes against the committed golden vectors (GPU box):
relative L2 / max error of latents, sample and taps for bf16 and f16 on every golden case, plus label agreement of
the full eval forward.  Output goes to stdout (copied into profiles/ and DESIGN.md)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import CASES, make_inputs, load_golden, tap_subset  # noqa: E402
from util import rel_err  # noqa: E402


def main():
    from madm_amd.ldm_rocm import LdmRocm
    m = LdmRocm("", [], [5, 8, 11], [], input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                compute_dtype=torch.float32, weights='synthetic', seed=0)
    m.vae_decoder_loss = True
    for name in ("small_t0", "small_t60", "rect_t0", "full_t0"):
        case, gold = CASES[name], load_golden(name)
        images, cond_inputs, cond_emb, timesteps, _ = make_inputs(**case)
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            m.compute_dtype = m.vae.compute_dtype = m.unet.compute_dtype = dt
            t = case["t"]
            feats, extra = m({"img": images.cuda(), "cond_inputs": cond_inputs.cuda(), "cond_emb": cond_emb.cuda(),
                              "timestep": (t, t + 1)}, "rgb", return_unet_final_output=True)
            torch.cuda.synchronize()
            ds = 4 if name.startswith("full") else 1
            row = [("latents", m.last_latents.cpu(), gold["latents"]),
                   ("sample", extra["before_vae.decoder"].cpu(), gold["sample"]),
                   ("decoder", feats[0].cpu()[:, :, ::ds, ::ds], gold["decoder"])]
            for i, f in enumerate(feats[1:]):
                row.append((f"tap{i}", tap_subset(name, f.cpu()), gold[f"tap{i}"]))
            print(f"{name:10s} {str(dt).split('.')[-1]:9s} " +
                  "  ".join(f"{k} l2 {rel_err(a, b)[1]:.2e} max {rel_err(a, b)[0]:.2e}" for k, a, b in row), flush=True)


if __name__ == "__main__":
    main()
