#!/bin/bash
# weight-streaming-bound 3x3 layers of the 8x8 / 16x16 UNet levels (29.5 MB of weights per launch): tile x split-K sweep
# with rotating (cold) weights inside a hipGraph.  usage (gpurun): bash tools/exp/wstream.sh
cd $GRAFT_REPO_ROOT
for sk in 12 16 24 32; do python tools/bench_one.py --hw 8 8 --cin 1280 --cout 1280 --tile 11 --splitk $sk --rotate 8 --graph --reps 40 --dtype f16 --no-stats 2>&1 | tail -1; done
for sk in 12 24; do python tools/bench_one.py --hw 8 8 --cin 1280 --cout 1280 --tile 7 --splitk $sk --rotate 8 --graph --reps 40 --dtype f16 --no-stats 2>&1 | tail -1; done
for sk in 12 24; do python tools/bench_one.py --hw 8 8 --cin 1280 --cout 1280 --tile 6 --splitk $sk --rotate 8 --graph --reps 40 --dtype f16 --no-stats 2>&1 | tail -1; done
echo "16x16 level"
for t in 10 11 7 8; do for sk in 3 6 12; do python tools/bench_one.py --hw 16 16 --cin 1280 --cout 1280 --tile $t --splitk $sk --rotate 8 --graph --reps 40 --dtype f16 --no-stats 2>&1 | tail -1; done; done
