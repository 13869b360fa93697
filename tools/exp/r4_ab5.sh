#!/bin/bash
# round 4: policy switches under the staged pipeline (same box, one call)
tag=${1:-r4ab5}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --steps 40 --warmup 8"
run() { name=$1; shift; env "$@" python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s value %7.1f img/s  step %6.3f ms  serial %6.3f ms' % ('$name', d['value'], d['ms_per_step'], d['serial_ms_per_step']))" | tee -a $O/ab.txt; }
run base X=1
run fuse_gn_max_n_320 MADM_FUSE_GN_MAX_N=320
run fuse_gn_max_n_640 MADM_FUSE_GN_MAX_N=640
run fuse_gn_max_n_1280 MADM_FUSE_GN_MAX_N=1280
run no_post_gn MADM_NO_POST_GN=1
run attn_nw8 MADM_ATTN_NW8=1
run attn_nq1 MADM_ATTN_NQ1=1
run base_again X=1
