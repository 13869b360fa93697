#!/bin/bash
# depthwise dilated conv on the dilation's lattice (LDS halo per residue class): correctness + A/B against the comb kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run25; mkdir -p $O
cd $R
python -m pytest tests/test_eval_gpu.py -q -m gpu -x -k "spatial or dwconv or misc" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python tools/exp/bench_dwconv.py 2>&1 | grep -v amdgpu.ids | tee $O/dwconv.txt
