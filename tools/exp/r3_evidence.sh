#!/bin/bash
# Round-3 experiment evidence in one call: GEMM microbenchmark, stamps of the A-stationary kernel and of the short-K LDS-DMA
# igemm (needs build/libmadm_APSTAMPS.so and build/libmadm_GLDSST.so, see the headers of tools/exp/stamps_*.py).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_evidence; mkdir -p $O; cd $R
{ echo "== default tiles (tuned table)"; python tools/bench_gemm.py; echo "== tile 13 forced where eligible"; python tools/bench_gemm.py --tile 13; } 2>&1 | grep -v amdgpu > $O/gemm_microbench.txt
if [ -f build/libmadm_APSTAMPS.so ]; then
  { for a in "8192 320 2560" "8192 320 960" "2048 640 5120"; do echo "== M K N = $a (GEGLU epilogue)"; MADM_HIP_LIB=build/libmadm_APSTAMPS.so python tools/exp/stamps_apanel.py $a; done; } 2>&1 | grep -v amdgpu > $O/apanel_stamps.txt
fi
if [ -f build/libmadm_GLDSST.so ]; then
  { for a in "64 320 320 1 11 1" "64 320 960 1 8 1" "32 640 640 1 11 1" "16 1280 1280 1 7 1"; do MADM_HIP_LIB=build/libmadm_GLDSST.so python tools/exp/stamps_glds.py $a; done; } 2>&1 | grep -v amdgpu > $O/glds_stamps_short_k.txt
fi
for f in $O/*.txt; do tail -n 3 $f; done
