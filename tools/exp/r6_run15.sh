#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6o; mkdir -p $O
cd $R
bash tools/first_touch.sh
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=30 > $O/pytest.log 2>&1; echo "suite rc=$? $(grep -E ' passed| failed' $O/pytest.log | tail -1)"
grep -A34 "slowest 30" $O/pytest.log | cut -c1-160
