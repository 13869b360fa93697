#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs (tools/pmc.sh) per kernel name.

  pmc_report.py <dir> [regex]            text view, mean counter value per dispatch
  pmc_report.py <dir> --json out.json    HBM traffic per launch keyed by bench.py's kernel names
                                         (FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads half of the
                                         streamed bytes -> x2, MI355X_MICROARCH.md "HBM / rocprofv3")"""
import collections
import csv
import glob
import json
import re
import sys


def bench_name(k):
    dt = lambda s: {"DF16b": "_bf16", "DF16_": "_f16"}.get(s, "_f32")
    m = re.search(r"conv3x3_h16_kernelI(DF16b|DF16_|f)Li(\d+)E", k)
    if m:
        return f"conv3x3_h16_x{m.group(2)}{dt(m.group(1))}"
    m = re.search(r"conv3x3_halo_(dma_)?kernelI(DF16b|DF16_|f)Li(\d+)E", k)
    if m:
        return f"conv3x3_halo_{m.group(1) or ''}x{m.group(3)}{dt(m.group(2))}"
    m = re.search(r"igemm_(glds_)?kernelI(DF16b|DF16_|f)Li(\d+)ELi(\d+)ELi(\d+)E", k)
    if m:
        deep = "d" if (not m.group(1) and m.group(3) == "64" and m.group(5) == "8") else ""
        return f"igemm_{m.group(1) or ''}{m.group(3)}x{m.group(4)}{deep}{dt(m.group(2))}"
    m = re.search(r"(splitk_reduce|gn_apply|gn_stats|layernorm|softmax_rows|image_to_im2col)_kernelI(DF16b|DF16_|f)", k)
    if m:
        return m.group(1) + dt(m.group(2))
    return None


def main():
    d = sys.argv[1]
    as_json = len(sys.argv) > 3 and sys.argv[2] == "--json"
    pat = "." if as_json or len(sys.argv) < 3 else sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = bench_name(k) if as_json else re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k)[:60]
            if k is None or not re.search(pat, k):
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    if as_json:
        # kernel wall time per launch from the kernel traces of the p5 pass (the pass the SQ co-execution counters come from)
        dur = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(d + "/p5/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = bench_name(r["Kernel_Name"])
                if k is not None:
                    dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                    dur[k][1] += 1
        out = {}
        for k, v in agg.items():
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                fetch = 2.0 * 1024.0 * v["FETCH_SIZE"] / cnt[k]["FETCH_SIZE"]
                write = 1024.0 * v["WRITE_SIZE"] / cnt[k]["WRITE_SIZE"]
                out[k] = {"hbm_bytes_per_launch": round(fetch + write), "fetch_bytes_per_launch": round(fetch),
                          "write_bytes_per_launch": round(write), "dispatches": cnt[k]["FETCH_SIZE"]}
                # per-SIMD cycle counts per launch (1 024 SIMDs): _MFMA_BUSY_ counts cycles, the others quad-cycles
                if "SQ_VALU_MFMA_COEXEC_CYCLES" in v:
                    per = lambda c, q: v[c] / cnt[k][c] * q / 1024.0
                    mf, va, co = per("SQ_VALU_MFMA_BUSY_CYCLES", 1), per("SQ_ACTIVE_INST_VALU", 4), per("SQ_VALU_MFMA_COEXEC_CYCLES", 4)
                    out[k]["sq"] = {"mfma_busy_cycles_per_simd": round(mf), "valu_active_cycles_per_simd": round(va),
                                    "coexec_cycles_per_simd": round(co), "coexec_share_of_valu": round(co / va, 3) if va else None,
                                    "wave_cycles_per_slot": round(per("SQ_WAVE_CYCLES", 4) / 2.0),
                                    "kernel_us": round(dur[k][0] / dur[k][1], 2) if dur[k][1] else None,
                                    "dispatches": cnt[k]["SQ_VALU_MFMA_COEXEC_CYCLES"]}
        json.dump({"source": "tools/pmc.sh (separate --pmc passes, --kernel-trace only); FETCH_SIZE KB x2 (gfx950), "
                             "WRITE_SIZE KB; mean over the dispatches of each kernel in the traced run",
                   "kernels": out}, open(sys.argv[3], "w"), indent=1)
        print(json.dumps(out, indent=1))
        return
    for k, v in agg.items():
        print(k)
        for c, val in sorted(v.items()):
            n = cnt[k][c]
            print(f"   {c:28s} {val / n:16.1f} per dispatch ({n} dispatches)")


if __name__ == "__main__":
    main()
