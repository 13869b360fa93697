#!/bin/bash
# The UNet's 3x3 convs (bs = 2) on the 16 x 16-patch kernel (tile 12) with split-K against the tuned 8 x 16-patch kernel
# (tile 10): fewer bytes per FLOP through the load path vs fewer, longer workgroups.  usage (through gpurun): bash tools/exp/h16_unet.sh
cd $GRAFT_REPO_ROOT
run() { python tools/bench_one.py --dtype f16 --graph --rotate 8 --reps 40 "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for shape in "64 64 320 320" "64 64 640 320" "32 32 640 640" "32 32 1280 640" "64 64 960 320"; do
  set -- $shape
  echo "== ${1}x${2} Cin $3 Cout $4"
  for cfg in "10 1" "10 2" "12 1" "12 2" "12 3" "12 5" "9 1" "9 2"; do
    set -- $shape $cfg
    echo -n "tile $5 sk $6: "; run --hw $1 $2 --cin $3 --cout $4 --tile $5 --splitk $6
  done
done
