#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6b; mkdir -p $O
cd $R
bash tools/first_touch.sh
timeout 1200 python -m pytest -q -x -p no:cacheprovider tests/test_host.py tests/test_poison_gpu.py::test_hbm_poison_harness_is_effective \
   "tests/test_train_gpu.py::test_teacher_side_stream_is_bit_identical_over_steps" \
   "tests/test_train_gpu.py::test_trainer_range_assert_fires_before_the_optimizer_step" \
   "tests/test_train_gpu.py::test_train_step_matches_fixture" \
   "tests/test_parity_gpu.py::test_fused_proj_out_matches_two_launch_path" -s > $O/new_tests.log 2>&1
echo "new tests rc=$? $(grep -E ' passed| failed' $O/new_tests.log | tail -1)"; grep -E "adapter tensors|^FAILED|Error" $O/new_tests.log | head -20
timeout 600 python -m pytest -q -x -p no:cacheprovider tests/test_parity_gpu.py::test_bench_workloads_have_tuned_rows > $O/tuned_rows.log 2>&1
echo "tuned rows rc=$?"; grep -E "^E  " $O/tuned_rows.log | head -80
SKIP_FIRST=1 bash tools/exp/r6_stagger.sh
timeout 600 python tools/exp/train_aten_sites.py > $O/train_aten_sites.txt 2>&1; head -50 $O/train_aten_sites.txt
