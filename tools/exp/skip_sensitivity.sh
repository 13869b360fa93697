#!/bin/bash
# What would the step cost without one kernel class?  (MADM_EXP_SKIP: launches skipped, results garbage, timing only.)
# Upper bounds for the gain of fusing / speeding up that class, pipelined (value) and serial (serial_ms_per_step).
# usage (through gpurun): bash tools/exp/skip_sensitivity.sh <tag>
tag=${1:-skip}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for s in none layernorm gn_apply attention tile12 tile10 "tile7,tile11" "tile8,tile2" "layernorm,gn_apply"; do
  if [ "$s" = none ]; then v=""; else v=$s; fi
  MADM_EXP_SKIP=$v python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype --steps 40 --warmup 8 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s value %7.1f img/s  step %6.3f ms  serial %6.3f ms' % ('$s', d['value'], d['ms_per_step'], d['serial_ms_per_step']))" | tee -a $O/skip.txt
done
