#!/bin/bash
tag=${1:-r4g9}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp/pkf32_repro.hip -o /tmp/pkf32_repro > $O/repro_build.log 2>&1
for cfg in "3000 1 1 0" "3000 1 1 1" "3000 0 1 0" "3000 1 0 0"; do timeout 60 /tmp/pkf32_repro $cfg; echo "exit $?"; done > $O/pkf32_repro.txt 2>&1; cat $O/pkf32_repro.txt | cut -c1-220
B="--no-cpu-baseline --no-alt-dtype --steps 40 --warmup 8"
python bench.py $B > $O/bench.json 2>/dev/null; python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","serial_ms_per_step","calib")})
print({n: v['ms'] for n, v in list(d['kernels'].items())[:8]})
PY
python bench.py --workload eval --steps 10 --warmup 2 2>/dev/null | cut -c1-200
python tools/tune_concurrent.py --workload eval --max-m 100000000 --min-us 20 --rows $O/tuned_eval.txt > $O/tune_eval.txt 2>&1; cat $O/tuned_eval.txt | head -40
MADM_TUNED_FILE=$O/tuned_eval.txt python bench.py --workload eval --steps 10 --warmup 2 2>/dev/null | cut -c1-200
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
