#!/bin/bash
tag=${1:-r4g16}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
t() { name=$1; shift; env "$@" python -m pytest tests/test_train_gpu.py -x -q -s -k "lora-f32" > $O/t_$name.log 2>&1; echo "$name rc=$? $(grep -o 'vae_decoder_target_loss [0-9./ ]*' $O/t_$name.log | head -1) $(grep -o 'near-tie pixels.*' $O/t_$name.log | head -1)"; }
t cur X=1
t cur_nofuse MADM_NO_FUSE_PROJ_OUT=1
t lib328e657_nofuse MADM_HIP_LIB=$R/build/libmadm_hip_328e657.so MADM_NO_FUSE_PROJ_OUT=1
t lib328e657 MADM_HIP_LIB=$R/build/libmadm_hip_328e657.so
t lib1ccdc76_nofuse MADM_HIP_LIB=$R/build/libmadm_hip_1ccdc76.so MADM_NO_FUSE_PROJ_OUT=1
t lib1ccdc76 MADM_HIP_LIB=$R/build/libmadm_hip_1ccdc76.so
python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | tail -2
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --steps 40 --warmup 8"
python bench.py $B 2>/dev/null | cut -c1-160
MADM_HIP_LIB=$R/build/libmadm_hip_oldattn.so python bench.py $B 2>/dev/null | cut -c1-160
python bench.py $B 2>/dev/null | cut -c1-160
