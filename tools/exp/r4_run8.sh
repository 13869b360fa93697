#!/bin/bash
tag=${1:-r4g8}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp/pkf32_repro.hip -o /tmp/pkf32_repro > $O/repro_build.log 2>&1; tail -2 $O/repro_build.log | cut -c1-200
for cfg in "20000 1 1 0" "20000 1 1 1" "20000 0 1 0" "20000 1 0 0"; do timeout 120 /tmp/pkf32_repro $cfg; done > $O/pkf32_repro.txt 2>&1; cat $O/pkf32_repro.txt | cut -c1-220
B="--no-cpu-baseline --no-alt-dtype --steps 40 --warmup 8"
run() { name=$1; shift; env "$@" python bench.py $B 2>/dev/null > $O/bench_$name.json; python -c "import sys,json; d=json.loads(open('$O/bench_$name.json').read()); print('%-30s value %7.1f img/s  step %6.3f ms  serial %6.3f ms' % ('$name', d['value'], d['ms_per_step'], d['serial_ms_per_step'])); k=d.get('kernels',{}); print('   ', {n: v['ms'] for n, v in list(k.items())[:12]})" | tee -a $O/ab.txt; }
run base X=1
run nopk MADM_HIP_LIB=$R/build/libmadm_hip_nopk.so
run base2 X=1
run nopk2 MADM_HIP_LIB=$R/build/libmadm_hip_nopk.so
MADM_HIP_LIB=$R/build/libmadm_hip_nopk.so python bench.py --workload eval --steps 10 --warmup 2 2>/dev/null | cut -c1-200
MADM_HIP_LIB=$R/build/libmadm_hip_nopk.so python bench.py --workload train --steps 5 --warmup 2 2>/dev/null | cut -c1-200
