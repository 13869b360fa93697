#!/bin/bash
# the driver's command line (K = 20, W = 5) against the default (K = 50, W = 10): what the pipeline's fill + drain cost a short timed region
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run41; mkdir -p $O
cd $R
for i in 1 2; do
 for kw in "20 5" "50 10" "100 10"; do set -- $kw
  python bench.py --gpus 1 --steps $1 --warmup $2 --no-cpu-baseline --no-kernel-profile --no-alt-dtype 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$1 W=$2', d['value'], d['ms_per_step'], d['serial_ms_per_step'], d['calib']['h16_128x128_512sq_us'])" | tee -a $O/k_sweep.txt
 done
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_like.json 2>/dev/null; cut -c1-200 $O/driver_like.json
