#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/qt_bench
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/qt_bench -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-profile --pipeline 3 --hw-queues ${1:-8} > gpurun_out/qt_bench.log 2>&1
f=$(find gpurun_out/qt_bench -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
g = collections.OrderedDict()
for r in rows:
    k = (r["Queue_Id"], r["Stream_Id"])
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if k not in g: g[k] = [0, s, e, 0]
    g[k][0] += 1; g[k][2] = max(g[k][2], e); g[k][3] += e - s
print("queue stream  dispatches  first_ms  last_ms  busy_ms")
for k, v in g.items(): print(k, v[0], round(v[1] / 1e6, 2), round(v[2] / 1e6, 2), round(v[3] / 1e6, 2))
PY
rm -rf gpurun_out/qt_bench
