#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O
cd $R
bash tools/first_touch.sh
(python tools/exp/train_determinism.py 12 f32 0
 python tools/exp/train_determinism.py 12 f32 1
 python tools/exp/train_determinism.py 8 f16 1) 2>&1 | grep -v amdgpu.ids | tee $O/train_determinism.txt
for i in 1 2 3; do
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_train_gpu.py -k "rccl_single_rank or teacher_side_stream or range_assert" > $O/train_tests_$i.log 2>&1
echo "train tests run $i rc=$? $(grep -E ' passed| failed' $O/train_tests_$i.log | tail -1)"; grep -E "^FAILED|^E  " $O/train_tests_$i.log | head -10
done
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_parity_gpu.py -k "test_golden or tuned_rows" -s > $O/parity.log 2>&1
echo "parity rc=$? $(grep -E ' passed| failed' $O/parity.log | tail -1)"; grep -E "^FAILED|^E  " $O/parity.log | head -20
timeout 600 python tools/exp/train_aten_sites.py > $O/train_aten_sites.txt 2>&1; head -60 $O/train_aten_sites.txt
python bench.py --workload train --steps 5 --warmup 2 2>/dev/null | cut -c1-600
MADM_NO_TEACHER_OVERLAP=1 python bench.py --workload train --steps 5 --warmup 2 2>/dev/null | cut -c1-600
