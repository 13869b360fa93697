#!/bin/bash
tag=${1:-r4g7}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
MADM_HIP_LIB=$R/build/libmadm_hip_pkcheck.so python tools/exp/pkf32_check.py --reps 20 > $O/pkcheck_2wg.txt 2>&1; grep -v "^   wg" $O/pkcheck_2wg.txt | cut -c1-200
MADM_HIP_LIB=$R/build/libmadm_hip_pknop.so python tools/exp/pkf32_check.py --reps 20 > $O/pkcheck_2wg_nop.txt 2>&1; grep -v "^   wg" $O/pkcheck_2wg_nop.txt | cut -c1-200
hipcc --offload-arch=gfx950 -O3 tools/exp/pkf32_repro.hip -o /tmp/pkf32_repro 2>/dev/null
for cfg in "20000 1 1 0" "20000 1 1 1" "20000 0 1 0" "20000 1 0 0"; do timeout 120 /tmp/pkf32_repro $cfg; done > $O/pkf32_repro.txt 2>&1; cat $O/pkf32_repro.txt | cut -c1-220
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","serial_ms_per_step","value_normalised","calib")})
PY
python bench.py --workload eval --steps 10 --warmup 2 > $O/bench_eval.json 2> $O/bench_eval.err; cut -c1-250 $O/bench_eval.json
python bench.py --workload slide --steps 5 --warmup 2 > $O/bench_slide.json 2> $O/bench_slide.err; cut -c1-250 $O/bench_slide.json
python bench.py --workload train --steps 5 --warmup 2 > $O/bench_train.json 2> $O/bench_train.err; cut -c1-250 $O/bench_train.json
