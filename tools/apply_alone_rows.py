#!/usr/bin/env python3
"""Writes madm_amd/csrc/igemm_tuned_latency.inc (the rows of the LATENCY profile, madm_set_tuning_profile(1)) from one or more
row files of tools/tune_concurrent.py --alone-rows; a later file overrides an earlier one on the same (dtype, M, N, K, KH, variant).
The train workload's where-clause of tune_concurrent.py (shapes of the extractor's table are skipped) does not apply here: the
latency table is consulted by synchronous callers only.   usage: apply_alone_rows.py rows1.txt [rows2.txt ...]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "madm_amd", "csrc", "igemm_tuned_latency.inc")
NEW = re.compile(r"^(\d+) (\d+) (\d+) (\d+) (\d+) (\d+) (\d+) (\d+)\s*#\s*(alone [\d.]+ -> [\d.]+ us.*)$")
HEAD = """// Latency-profile rows (madm_set_tuning_profile(1)): {dtype, M, N, K, KH, variant, tile, splitk} as in igemm_tuned.inc, chosen by the
// time of ONE launch on an idle chip (tools/tune_concurrent.py --alone-rows; tools/apply_alone_rows.py).  Consulted in front of
// igemm_tuned.inc by a synchronous forward() (one batch in flight); the throughput runners never see them.
"""


def main():
    rows = {}
    for path in sys.argv[1:]:
        tag = os.path.basename(path)
        for line in open(path):
            m = NEW.match(line.strip())
            if m:
                key = tuple(int(m.group(i)) for i in range(1, 7))
                rows[key] = (int(m.group(7)), int(m.group(8)), f"{m.group(9)} [{tag}]")
    # a "winner" that IS the throughput table's row is measurement noise (the same configuration timed twice): dropped
    base = {}
    for line in open(os.path.join(ROOT, "madm_amd", "csrc", "igemm_tuned.inc")):
        m = re.match(r"^\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\}", line)
        if m:
            v = tuple(int(x) for x in m.groups())
            base[v[:6]] = v[6:]
    same = [k for k, r in rows.items() if base.get(k) == (r[0], r[1])]
    for k in same:
        del rows[k]
    print(f"{len(same)} rows equal to the throughput table's dropped")
    with open(INC, "w") as f:
        f.write(HEAD)
        for k in sorted(rows):
            t, sk, why = rows[k]
            f.write("{%d, %d, %d, %d, %d, %d, %d, %d},   // %s\n" % (*k, t, sk, why))
    print(f"{len(rows)} rows written to {INC}")


if __name__ == "__main__":
    main()
