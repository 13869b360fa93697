#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6g; mkdir -p $O
cd $R
bash tools/first_touch.sh
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_train_gpu.py -k "rccl_single_rank or teacher_side_stream or range_assert" -s > $O/train_tests.log 2>&1
echo "train tests rc=$? $(grep -E ' passed| failed' $O/train_tests.log | tail -1)"; grep -E "^FAILED|^E  |   step " $O/train_tests.log | cut -c1-200 | head -60
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "suite rc=$? $(grep -E ' passed| failed' $O/pytest.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.log | head -30
