#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6h; mkdir -p $O
cd $R
bash tools/first_touch.sh
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_eval_gpu.py -k "staged_inference or pipelined_matches" tests/test_train_gpu.py::test_teacher_side_stream_is_bit_identical_over_steps > $O/tests.log 2>&1
echo "tests rc=$? $(grep -E ' passed| failed' $O/tests.log | tail -1)"; grep -E "^FAILED|^E  " $O/tests.log | cut -c1-300 | head -20
show() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], 'images/s', d['ms_per_step'], 'ms/step; serial', d.get('serial_ms_per_step'))"; }
for wl in eval slide; do
  python bench.py --workload $wl --steps 40 --warmup 8 --no-kernel-profile 2>$O/err_$wl.txt | show "$wl graphed 4 streams"
  for cfg in "2 6" "2 8" "1 4" "2 4"; do
    set -- $cfg
    python bench.py --workload $wl --steps 40 --warmup 8 --no-kernel-profile --eval-runner staged --pipeline $1 --slots $2 2>>$O/err_$wl.txt | show "$wl staged unet_streams $1 slots $2"
  done
done | tee $O/eval_ab.txt
tail -5 $O/err_eval.txt
