#!/usr/bin/env python3
"""Rebuilds madm_amd/csrc/igemm_tuned.inc from the outputs of tools/tune_insitu.py.
usage: apply_tuned.py <extract.txt> [<eval.txt>]   (a missing eval file keeps the eval rows of the current table
that do not collide with an extract shape)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "madm_amd", "csrc", "igemm_tuned.inc")
ROW = re.compile(r"^\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},")


LINE = re.compile(r"^M(\d+) N(\d+) K(\d+) k(\d+) s\d+((?: up)?(?: cat)?(?: st)?(?: gn)?)\s+\d+\s+t\d+/sk\d+\s+[\d.]+\s+t(\d+)/sk(\d+)")


def rows_of(path):
    """Rows from the tuner's per-shape table (dtype bf16): the first (most expensive) occurrence of a key wins."""
    out, seen = [], set()
    for line in open(path):
        m = LINE.match(line.strip())
        if not m:
            continue
        M, N, K, KH = (int(m.group(i)) for i in range(1, 5))
        variant = 1 if " gn" in m.group(5) else (2 if " up" in m.group(5) else 0)
        key = (1, M, N, K, KH, variant)
        if key in seen:
            continue
        seen.add(key)
        tile = int(m.group(6))
        out.append(key + (tile, int(m.group(7))))
    return sorted(out)


def main():
    """apply_tuned.py <extract.txt>[,<extract2.txt>...] [<eval.txt>[,<eval2.txt>...]]
    Several tuner outputs (e.g. tuned under different GroupNorm-fusion policies, so that both the plain and the fused
    variant of a shape are covered) are merged; for a key present in several files the FIRST file wins."""
    def merged(arg):
        out, seen = [], set()
        for path in arg.split(","):
            for r in rows_of(path):
                if r[:6] not in seen:
                    seen.add(r[:6])
                    out.append(r)
        return sorted(out)

    extract = merged(sys.argv[1])
    if len(sys.argv) > 2:
        ev = merged(sys.argv[2])
    else:
        ev, on = [], False
        for line in open(INC):
            if "full inference forward" in line:
                on = True
            m = ROW.match(line.strip())
            if on and m:
                ev.append(tuple(int(x) for x in m.groups()))
    keys = {r[:6] for r in extract}
    ev = [r for r in ev if r[:6] not in keys]
    with open(INC, "w") as f:
        f.write("// {dtype (0 f32, 1 bf16), M, N, K, KH, variant (0 plain, 1 GroupNorm-fused, 2 upsample), tile (1=128x128, "
                "2=128x64, 3=64x64 igemm; 4 / 5 = halo conv3x3 x128 / x64; 6 = 64x64 igemm, 8-deep prefetch; 7 / 8 / 11 = "
                "LDS-DMA igemm 64x64 / 128x64 / 64x64 short ring; 9 / 10 = LDS-DMA halo x128 / x64), splitk}\n")
        f.write("// measured by tools/tune_insitu.py on MI355X (whole eager forwards, cold weights), bf16, round 1; plain and\n"
                "// GroupNorm-fused variants come from runs under different fusion policies (MADM_FUSE_GN_MAX_N)\n")
        f.write("// -- feature extractor, bs=2, 512x512 (BASELINE configs[1])\n")
        for r in extract:
            f.write("{%d, %d, %d, %d, %d, %d, %d, %d},\n" % r)
        f.write("// -- full inference forward, bs=1, 512x512, RGB->Depth config (BASELINE configs[2]): additional shapes\n")
        for r in ev:
            f.write("{%d, %d, %d, %d, %d, %d, %d, %d},\n" % r)
    print(f"{len(extract)} extract rows, {len(ev)} eval rows -> {INC}")


if __name__ == "__main__":
    main()
