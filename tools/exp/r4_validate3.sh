#!/bin/bash
# round 4: bench line (calib fields) + the whole GPU suite with its wall time
tag=${1:-r4v3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","serial_ms_per_step","value_normalised","calib","whole_path_roofline_frac")})
print(d.get("roofline"))
for k,v in list(d.get("kernels",{}).items())[:14]: print(k,v)
PY
( time python -m pytest tests -m gpu -x -q --durations=30 ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -45 $O/pytest.log
