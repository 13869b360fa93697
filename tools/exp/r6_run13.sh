#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6m; mkdir -p $O
cd $R
bash tools/first_touch.sh
for shape in "512 512 128 128 --gn --residual" "256 256 256 256 --gn --residual" "128 128 512 512 --residual" "64 64 512 512 --residual"; do
  set -- $shape
  for lib in default h16respf; do
    echo -n "$lib: "
    if [ $lib = default ]; then unset MADM_HIP_LIB; else export MADM_HIP_LIB=$R/build/libmadm_hip_$lib.so; fi
    python tools/bench_one.py --hw $1 $2 --cin $3 --cout $4 ${@:5} --tile 12 --dtype f16 --graph --reps 20 --rotate 4 --check 9 2>&1 | grep -E "TF/s|check" | tr '\n' ' '; echo
  done
done | tee $O/res_prefetch_layers.txt
unset MADM_HIP_LIB
show() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['unit'], d['ms_per_step'], 'ms/step; serial', d.get('serial_ms_per_step'))"; }
for i in 1 2; do
python bench.py --no-cpu-baseline --no-alt-dtype --no-kernel-profile 2>/dev/null | show "default                       "
MADM_HIP_LIB=$R/build/libmadm_hip_h16respf.so python bench.py --no-cpu-baseline --no-alt-dtype --no-kernel-profile 2>/dev/null | show "residual L2 prefetch          "
MADM_TUNED_FILE=$R/tools/exp/r6_side_rows_w0.txt python bench.py --no-cpu-baseline --no-alt-dtype --no-kernel-profile 2>/dev/null | show "side-by-side rows at weight 0 "
done | tee $O/bench_ab.txt
