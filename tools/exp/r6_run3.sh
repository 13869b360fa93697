#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c; mkdir -p $O
cd $R
bash tools/first_touch.sh
timeout 1500 python -m pytest -q -x -p no:cacheprovider \
   "tests/test_train_gpu.py::test_teacher_side_stream_is_bit_identical_over_steps" \
   "tests/test_train_gpu.py::test_trainer_range_assert_fires_before_the_optimizer_step" \
   "tests/test_train_gpu.py::test_train_step_matches_fixture" \
   "tests/test_parity_gpu.py::test_fused_proj_out_matches_two_launch_path" tests/test_poison_gpu.py::test_hbm_poison_harness_is_effective \
   tests/test_eval_gpu.py tests/test_ops_gpu.py -s > $O/new_tests.log 2>&1
echo "new tests rc=$? $(grep -E ' passed| failed' $O/new_tests.log | tail -1)"; grep -E "adapter tensors|^FAILED|Error" $O/new_tests.log | head -20
export MADM_HIP_LIB=$R/build/libmadm_hip_h16stamps.so
(python tools/exp/stamps_h16_rt.py 128 128 512 0 0
 python tools/exp/stamps_h16_rt.py 128 128 512 1 0
 python tools/exp/stamps_h16_rt.py 128 128 512 1 1
 B=1 python tools/exp/stamps_h16_rt.py 128 128 512 0 0
 python tools/exp/stamps_h16_rt.py 256 256 256 1 0
 python tools/exp/stamps_h16_rt.py 512 512 128 1 0) 2>&1 | grep -v amdgpu.ids | tee $O/stamps_rt.txt
unset MADM_HIP_LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "^\s*(Name|Counter).*(COEXEC|MFMA|TRANS|VALU|BUSY|WAIT|LDS)" | sort -u | head -80 > $O/counters.txt
wc -l $O/counters.txt
for set in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
 for v in "" "--gn"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' _)$v
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -- python3 $R/tools/bench_one.py --hw 512 512 --cin 128 --cout 128 $v --tile 12 --dtype f16 --reps 5 > $O/pmc_$tag.log 2>&1
 done
done
cd $R; for d in $O/pmc_*/; do echo "== $d"; python tools/pmc_report.py $d h16; done > $O/pmc_report.txt 2>&1; ls $O
