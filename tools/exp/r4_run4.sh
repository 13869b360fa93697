#!/bin/bash
tag=${1:-r4g4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py -x -q -k "glds128sq" > $O/pytest_tiles.log 2>&1; echo "pytest new tiles rc=$?"; tail -3 $O/pytest_tiles.log
python -m pytest tests/test_train_gpu.py -x -q -k "rccl" > $O/pytest_rccl.log 2>&1; echo "pytest rccl rc=$?"; tail -3 $O/pytest_rccl.log
MADM_HIP_LIB=$R/build/libmadm_hip_pkcheck.so python tools/exp/pkf32_check.py --reps 30 > $O/pkcheck_2wg.txt 2>&1; cat $O/pkcheck_2wg.txt | cut -c1-330
MADM_APANEL_LDS_PAD=90000 MADM_HIP_LIB=$R/build/libmadm_hip_pkcheck.so python tools/exp/pkf32_check.py --reps 30 > $O/pkcheck_1wg.txt 2>&1; grep "launches" $O/pkcheck_1wg.txt
python tools/tune_concurrent.py --max-m 100000000 --min-us 15 --rows $O/tuned_side.txt > $O/tune_concurrent.txt 2>&1; grep -c . $O/tuned_side.txt; grep "t14\|t15" $O/tune_concurrent.txt | head -40
bash tools/exp/r4_skip.sh $tag
