#!/bin/bash
tag=${1:-r4g13}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py -x -q -k "conv1x1_fused_groupnorm or conv_fused_groupnorm or test_conv2d" > $O/pytest_ops.log 2>&1; echo "pytest ops rc=$?"; tail -4 $O/pytest_ops.log
python -m pytest tests/test_parity_gpu.py tests/test_train_gpu.py tests/test_eval_gpu.py -x -q -k "golden or staged or fixture" > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -4 $O/pytest_parity.log
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --steps 40 --warmup 8"
run() { name=$1; shift; env "$@" python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-30s value %7.1f img/s  step %6.3f ms  serial %6.3f ms' % ('$name', d['value'], d['ms_per_step'], d['serial_ms_per_step']))" | tee -a $O/ab.txt; }
run new X=1
run no_gn1x1 MADM_NO_FUSE_GN_1X1=1
run new2 X=1
run no_gn1x1_2 MADM_NO_FUSE_GN_1X1=1
