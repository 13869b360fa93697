#!/bin/bash
tag=${1:-r4g17}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
t() { name=$1; shift; env "$@" python -m pytest tests/test_train_gpu.py -x -q -s -k "lora-f32" > $O/t_$name.log 2>&1; echo "$name rc=$? $(grep -o 'vae_decoder_target_loss [0-9./ ]*' $O/t_$name.log | head -1) $(grep -o 'near-tie pixels: [0-9].*' $O/t_$name.log | head -1)"; }
t cur X=1
t hybA MADM_HIP_LIB=$R/build/libmadm_hip_hybA.so
t hybB MADM_HIP_LIB=$R/build/libmadm_hip_hybB.so
t hybC MADM_HIP_LIB=$R/build/libmadm_hip_hybC.so
