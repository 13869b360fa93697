#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_benches; mkdir -p $O; cd $R
python bench.py --workload eval --steps 10 --warmup 3 --no-cpu-baseline > $O/eval.json 2>$O/eval.err; cut -c1-220 $O/eval.json
MADM_NO_H16=1 python bench.py --workload eval --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > $O/eval_noh16.json 2>/dev/null; cut -c1-220 $O/eval_noh16.json
python bench.py --workload slide --steps 5 --warmup 2 --no-cpu-baseline > $O/slide.json 2>$O/slide.err; cut -c1-260 $O/slide.json; tail -3 $O/slide.err
python bench.py --workload train --steps 5 --warmup 2 > $O/train.json 2>$O/train.err; cut -c1-200 $O/train.json; tail -3 $O/train.err
