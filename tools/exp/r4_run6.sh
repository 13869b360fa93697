#!/bin/bash
tag=${1:-r4g6}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python -m pytest tests/test_parity_gpu.py -x -q -k "golden or staged" > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -3 $O/pytest_parity.log
python -m pytest tests/test_train_gpu.py -x -q -k "rccl or depth-f32" > $O/pytest_rccl.log 2>&1; echo "pytest rccl+train rc=$?"; tail -3 $O/pytest_rccl.log
MADM_HIP_LIB=$R/build/libmadm_hip_pkcheck.so python tools/exp/pkf32_check.py --reps 30 > $O/pkcheck_2wg.txt 2>&1; grep -v "^   wg" $O/pkcheck_2wg.txt | cut -c1-200
MADM_APANEL_LDS_PAD=60000 MADM_HIP_LIB=$R/build/libmadm_hip_pkcheck.so python tools/exp/pkf32_check.py --reps 30 > $O/pkcheck_1wg.txt 2>&1; grep "launches\|Error" $O/pkcheck_1wg.txt
python tools/tune_concurrent.py --max-m 100000000 --min-us 15 --only "K(1600|3200|6400) k1|k3 s2" --rows $O/tuned_side.txt > $O/tune_concurrent.txt 2>&1; cat $O/tuned_side.txt
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --steps 40 --warmup 8"
run() { name=$1; shift; env "$@" python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s value %7.1f img/s  step %6.3f ms  serial %6.3f ms' % ('$name', d['value'], d['ms_per_step'], d['serial_ms_per_step']))" | tee -a $O/ab.txt; }
run base X=1
run base_tuned MADM_TUNED_FILE=$O/tuned_side.txt
run no_fuse_proj_out MADM_NO_FUSE_PROJ_OUT=1
run fuse_gn_max_n_320 MADM_FUSE_GN_MAX_N=320 MADM_TUNED_FILE=$O/tuned_side.txt
run fuse_gn_max_n_640 MADM_FUSE_GN_MAX_N=640 MADM_TUNED_FILE=$O/tuned_side.txt
run no_post_gn MADM_NO_POST_GN=1 MADM_TUNED_FILE=$O/tuned_side.txt
run attn_nw8 MADM_ATTN_NW8=1 MADM_TUNED_FILE=$O/tuned_side.txt
run attn_nq1 MADM_ATTN_NQ1=1 MADM_TUNED_FILE=$O/tuned_side.txt
run base_tuned_again MADM_TUNED_FILE=$O/tuned_side.txt
