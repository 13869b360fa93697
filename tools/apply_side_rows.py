#!/usr/bin/env python3
"""Merges rows of tools/tune_concurrent.py (MADM_TUNED_FILE format, with its '# side a -> b us, alone c -> d' comments) into
madm_amd/csrc/igemm_tuned.inc: a row replaces the table's row of the same (dtype, M, N, K, KH, variant) or is appended.
Accepted when  side + W * alone  improves by more than 3 %  (W = 0.25: the pipelined headline counts what a launch takes out
of the chip, the serial step its latency; an 8 % better side time is not bought with a 30 % longer latency).
usage: apply_side_rows.py <rows.txt> [--weight 0.25] [--force "M N K KH variant tile splitk" ...]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "madm_amd", "csrc", "igemm_tuned.inc")
ROW = re.compile(r"^\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},(.*)$")
NEW = re.compile(r"^(\d+) (\d+) (\d+) (\d+) (\d+) (\d+) (\d+) (\d+)\s*#\s*side ([\d.]+) -> ([\d.]+) us, alone ([\d.]+) -> ([\d.]+)")


def main():
    w = 0.25
    if "--weight" in sys.argv:
        w = float(sys.argv[sys.argv.index("--weight") + 1])
    new = {}
    for line in open(sys.argv[1]):
        m = NEW.match(line.strip())
        if not m:
            continue
        key = tuple(int(m.group(i)) for i in range(1, 7))
        tile, sk = int(m.group(7)), int(m.group(8))
        s0, s1, a0, a1 = (float(m.group(i)) for i in range(9, 13))
        if s1 + w * a1 < 0.97 * (s0 + w * a0):
            new[key] = (tile, sk, f"r5 side-by-side: side {s0:.1f} -> {s1:.1f} us, alone {a0:.1f} -> {a1:.1f}")
        else:
            print("rejected", key, tile, sk, f"side {s0} -> {s1}, alone {a0} -> {a1}")
    lines = open(INC).read().split("\n")
    out, used = [], set()
    for ln in lines:
        m = ROW.match(ln)
        if m:
            key = tuple(int(m.group(i)) for i in range(1, 7))
            if key in new:
                t, sk, why = new[key]
                old_t, old_sk = int(m.group(7)), int(m.group(8))
                if (t, sk) != (old_t, old_sk):
                    ln = "{%d, %d, %d, %d, %d, %d, %d, %d},   // %s (was tile %d sk%d)" % (*key, t, sk, why, old_t, old_sk)
                used.add(key)
        out.append(ln)
    extra = [k for k in new if k not in used]
    if extra:
        while out and out[-1] == "":
            out.pop()
        out.append("// -- round 5: shapes without a row so far (tools/tune_concurrent.py; variant 3 = stride 2)")
        for k in sorted(extra):
            t, sk, why = new[k]
            out.append("{%d, %d, %d, %d, %d, %d, %d, %d},   // %s" % (*k, t, sk, why))
        out.append("")
    open(INC, "w").write("\n".join(out))
    print(f"{len(used)} rows replaced, {len(extra)} appended")


if __name__ == "__main__":
    main()
