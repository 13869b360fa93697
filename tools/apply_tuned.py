#!/usr/bin/env python3
"""Rebuilds madm_amd/csrc/igemm_tuned.inc from the outputs of tools/tune_insitu.py.
usage: apply_tuned.py <extract.txt> [<eval.txt>]   (a missing eval file keeps the eval rows of the current table
that do not collide with an extract shape)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "madm_amd", "csrc", "igemm_tuned.inc")
ROW = re.compile(r"^\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},")


def rows_of(path):
    out, on = [], False
    for line in open(path):
        if "rows for igemm_tuned.inc" in line:
            on = True
            continue
        m = ROW.match(line.strip())
        if on and m:
            out.append(tuple(int(x) for x in m.groups()))
    return out


def main():
    extract = rows_of(sys.argv[1])
    if len(sys.argv) > 2:
        ev = rows_of(sys.argv[2])
    else:
        ev, on = [], False
        for line in open(INC):
            if "full inference forward" in line:
                on = True
            m = ROW.match(line.strip())
            if on and m:
                ev.append(tuple(int(x) for x in m.groups()))
    keys = {r[:5] for r in extract}
    ev = [r for r in ev if r[:5] not in keys]
    with open(INC, "w") as f:
        f.write("// {dtype (0 f32, 1 bf16), M, N, K, KH, tile (1=128x128, 2=128x64, 3=64x64 igemm; 4 = halo conv3x3 x128, "
                "5 = halo x64; 6 = 64x64 igemm with the 8-deep prefetch), splitk}\n")
        f.write("// measured by tools/tune_insitu.py on MI355X (whole eager forwards, cold weights), bf16, round 1\n")
        f.write("// -- feature extractor, bs=2, 512x512 (BASELINE configs[1])\n")
        for r in extract:
            f.write("{%d, %d, %d, %d, %d, %d, %d},\n" % r)
        f.write("// -- full inference forward, bs=1, 512x512, RGB->Depth config (BASELINE configs[2]): additional shapes\n")
        for r in ev:
            f.write("{%d, %d, %d, %d, %d, %d, %d},\n" % r)
    print(f"{len(extract)} extract rows, {len(ev)} eval rows -> {INC}")


if __name__ == "__main__":
    main()
