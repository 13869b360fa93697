python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -2
B="python tools/bench_one.py --graph --reps 32"
$B --tile 9 --gn 2>&1 | tail -1; $B --tile 9 2>&1 | tail -1; $B --tile 4 --gn 2>&1 | tail -1
$B --hw 256 256 --cin 256 --cout 256 --tile 9 --gn 2>&1 | tail -1
$B --hw 64 64 --cin 320 --cout 320 --k 1 --tile 3 --reps 128 --rotate 16 2>&1 | tail -1
$B --hw 64 64 --cin 320 --cout 320 --k 1 --tile 7 --reps 128 --rotate 16 2>&1 | tail -1
$B --hw 16 16 --cin 1280 --cout 1280 --k 1 --tile 7 --reps 128 --rotate 16 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-kernel-profile --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['serial_images_per_s_per_gpu'])"
