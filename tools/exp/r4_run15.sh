#!/bin/bash
tag=${1:-r4g15}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest attention rc=$?"; tail -3 $O/pytest_attn.log
MADM_HIP_LIB=$R/build/libmadm_hip_attnstamps.so python tools/exp/stamps_attn.py 4096 40 > $O/stamps_attn_new.txt 2>&1; tail -4 $O/stamps_attn_new.txt
python -m pytest tests/test_parity_gpu.py tests/test_train_gpu.py -x -q -k "golden or staged or fixture" > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -4 $O/pytest_parity.log
B="--no-cpu-baseline --no-alt-dtype --steps 40 --warmup 8"
run() { name=$1; shift; env "$@" python bench.py $B 2>/dev/null > $O/bench_$name.json; python -c "import sys,json; d=json.loads(open('$O/bench_$name.json').read()); k=d.get('kernels',{}); print('%-30s value %7.1f img/s  step %6.3f ms  serial %6.3f ms  attn_d40 %s d80 %s d160 %s' % ('$name', d['value'], d['ms_per_step'], d['serial_ms_per_step'], k.get('attn_d40_f16',{}).get('ms'), k.get('attn_d80_f16',{}).get('ms'), k.get('attn_d160_f16',{}).get('ms')))" | tee -a $O/ab.txt; }
run new X=1
run oldattn MADM_HIP_LIB=$R/build/libmadm_hip_oldattn.so
run new2 X=1
run oldattn2 MADM_HIP_LIB=$R/build/libmadm_hip_oldattn.so
