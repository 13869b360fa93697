#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6i; mkdir -p $O
cd $R
[ -z "$SKIP_FIRST" ] && bash tools/first_touch.sh
for i in 1 2 3; do
  for m in 0 1; do
    echo -n "MADM_NO_TEACHER_OVERLAP=$m run $i: "
    MADM_NO_TEACHER_OVERLAP=$m python bench.py --workload train --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms/step')"
  done
done | tee $O/train_overlap_ab.txt
timeout 600 python tools/exp/train_aten_sites.py > $O/train_aten_sites.txt 2>&1; grep -v "amdgpu.ids\|Warning\|_warn" $O/train_aten_sites.txt | head -70
