#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6d; mkdir -p $O
cd $R
bash tools/first_touch.sh
for shape in "512 512 128 128" "512 512 128 128 --gn" "512 512 128 128 --gn --residual" "256 256 256 256 --gn" "128 128 512 512 --gn" "256 256 128 256" "64 64 320 320"; do
  set -- $shape
  for pr in 0 1 2 3; do
    echo -n "tap prio $pr: "
    MADM_H16_TAP_PRIO=$pr python tools/bench_one.py --hw $1 $2 --cin $3 --cout $4 ${@:5} --tile 12 --dtype f16 --graph --reps 20 --rotate 4 2>&1 | grep "TF/s"
  done
done | tee $O/layers_prio.txt
export MADM_HIP_LIB=$R/build/libmadm_hip_h16stamps.so
(MADM_H16_TAP_PRIO=1 python tools/exp/stamps_h16_rt.py 128 128 512 0 0
 MADM_H16_TAP_PRIO=1 python tools/exp/stamps_h16_rt.py 128 128 512 1 0) 2>&1 | grep -v amdgpu.ids | tee $O/stamps_rt_prio1.txt
unset MADM_HIP_LIB
for pr in 0 1 2; do
  echo "== bench, tap prio $pr"
  MADM_H16_TAP_PRIO=$pr python bench.py --no-cpu-baseline --no-alt-dtype 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'serial', d.get('serial_ms_per_step'), 'roofline', d['roofline']['achieved'], d['roofline']['frac'])"
done | tee $O/bench_prio.txt
timeout 1500 python -m pytest -q -x -p no:cacheprovider \
   "tests/test_train_gpu.py::test_teacher_side_stream_is_bit_identical_over_steps" \
   "tests/test_train_gpu.py::test_trainer_range_assert_fires_before_the_optimizer_step" \
   "tests/test_train_gpu.py::test_train_step_matches_fixture" \
   "tests/test_parity_gpu.py::test_fused_proj_out_matches_two_launch_path" tests/test_poison_gpu.py::test_hbm_poison_harness_is_effective \
   tests/test_parity_gpu.py::test_bench_workloads_have_tuned_rows tests/test_eval_gpu.py tests/test_ops_gpu.py -s > $O/new_tests.log 2>&1
echo "new tests rc=$? $(grep -E ' passed| failed' $O/new_tests.log | tail -1)"; grep -E "adapter tensors|^FAILED|Error" $O/new_tests.log | head -20
