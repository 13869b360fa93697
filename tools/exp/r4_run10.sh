#!/bin/bash
tag=${1:-r4g10}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest attention rc=$?"; tail -3 $O/pytest_attn.log
python -m pytest tests/test_parity_gpu.py tests/test_train_gpu.py -x -q -k "golden or staged or lora-f32" > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -3 $O/pytest_parity.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp/pkf32_repro.hip -o /tmp/pkf32_repro > $O/repro_build.log 2>&1
for cfg in "3000 1 1 0" "3000 1 0 0"; do timeout 60 /tmp/pkf32_repro $cfg; done > $O/pkf32_repro.txt 2>&1; grep -v "^   wg" $O/pkf32_repro.txt | cut -c1-200
B="--no-cpu-baseline --no-alt-dtype --steps 40 --warmup 8"
run() { name=$1; shift; env "$@" python bench.py $B 2>/dev/null > $O/bench_$name.json; python -c "import sys,json; d=json.loads(open('$O/bench_$name.json').read()); k=d.get('kernels',{}); print('%-30s value %7.1f img/s  step %6.3f ms  serial %6.3f ms  attn_d40 %s' % ('$name', d['value'], d['ms_per_step'], d['serial_ms_per_step'], k.get('attn_d40_f16',{}).get('ms')))" | tee -a $O/ab.txt; }
run new X=1
run oldattn MADM_HIP_LIB=$R/build/libmadm_hip_oldattn.so
run new2 X=1
run oldattn2 MADM_HIP_LIB=$R/build/libmadm_hip_oldattn.so
run cumask_enc0-160_unet96-256 MADM_EXP_CUMASK=0-160,96-256
run cumask_enc0-128_unet128-256 MADM_EXP_CUMASK=0-128,128-256
run cumask_enc0-256_unet64-256 MADM_EXP_CUMASK=0-256,64-256
run cumask_enc0-192_unet0-256 MADM_EXP_CUMASK=0-192,0-256
run new3 X=1
python tools/tune_concurrent.py --batch 3 --max-m 100000000 --min-us 25 --rows $O/tuned_b3.txt > $O/tune_b3.txt 2>&1; grep -c . $O/tuned_b3.txt
python bench.py --workload slide --steps 5 --warmup 2 2>/dev/null | cut -c1-200
MADM_TUNED_FILE=$O/tuned_b3.txt python bench.py --workload slide --steps 5 --warmup 2 2>/dev/null | cut -c1-200
