#!/bin/bash
tag=${1:-r4g14}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
MADM_HIP_LIB=$R/build/libmadm_hip_attnstamps.so python tools/exp/stamps_attn.py 4096 40 > $O/stamps_attn_new.txt 2>&1; tail -9 $O/stamps_attn_new.txt
MADM_HIP_LIB=$R/build/libmadm_hip_oldattnstamps.so python tools/exp/stamps_attn.py 4096 40 > $O/stamps_attn_old.txt 2>&1; tail -3 $O/stamps_attn_old.txt
python -m pytest tests/test_parity_gpu.py tests/test_train_gpu.py -x -q -k "golden or staged or fixture" > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -4 $O/pytest_parity.log
