#!/bin/bash
# round 4: what each kernel class costs in the pipelined step with the side-by-side table (MADM_EXP_SKIP, timing only)
tag=${1:-r4skip}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for s in none layernorm gn_apply attention "re:attn d40 Lq4096 Lk4096" "re:^k3 s1 M(524288|131072|32768) " "re:^k3 s2" "re:^k3 s1 M8192 N512" \
         "re:^k3 s1( up)? M(8192 N(320|640)|2048|512|128) " "re:^k1 s1 M(8192|2048|512|128) N(320|640|1280) K(320|640|1280)$" \
         "re:^k1 s1 M(8192 N2560|2048 N5120|512 N10240)" "re:^k1 s1 M(8192 N320 K1600|2048 N640 K3200|512 N1280 K6400)" \
         "re:^k1 s1 M(8192 N960|2048 N1920|512 N3840)" "re:^k1 s1 M(131072|32768|4096)"; do
  if [ "$s" = none ]; then v=""; else v=$s; fi
  MADM_EXP_SKIP="$v" python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype --steps 40 --warmup 8 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-70s value %7.1f img/s  step %6.3f ms  serial %6.3f ms' % ('''$s''', d['value'], d['ms_per_step'], d['serial_ms_per_step']))" | tee -a $O/skip.txt
done
