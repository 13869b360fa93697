#!/bin/bash
# every kernel launch preceded by an LDS poison (quiet NaNs in all LDS of the chip): does any kernel read LDS it did not write?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5r; mkdir -p $O
cd $R
export MADM_DEBUG_POISON_LDS=1
python -m pytest tests/test_ops_gpu.py tests/test_labels_gpu.py -q > $O/poison_ops.log 2>&1; echo "ops rc=$?"; tail -4 $O/poison_ops.log
python -m pytest tests/test_eval_gpu.py tests/test_parity_gpu.py -q > $O/poison_e2e.log 2>&1; echo "e2e rc=$?"; tail -4 $O/poison_e2e.log
python -m pytest tests/test_train_gpu.py -q -k "matches_fixture and (depth-f32 or depth-f16 or lora-f32) or attention_backward or unet_backward" > $O/poison_train.log 2>&1; echo "train rc=$?"; tail -4 $O/poison_train.log
