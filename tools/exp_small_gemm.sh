python -m pytest tests/test_ops_gpu.py tests/test_parity_gpu.py -x -q 2>&1 | tail -2
python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved']); print({k:(v['launches'],v['ms']) for k,v in d['kernels'].items()})"
for sk in 1 2 4; do for t in 3 2; do python tools/bench_one.py --hw 16 16 --cin 1280 --cout 1280 --k 1 --tile $t --splitk $sk --rotate 64 --reps 200 --no-stats; done; done
python tools/bench_one.py --hw 16 16 --cin 1280 --cout 1280 --k 1 --tile 3 --splitk 1 --rotate 1 --reps 200 --no-stats
for sk in 1 2 4; do python tools/bench_one.py --hw 32 32 --cin 640 --cout 640 --k 1 --tile 3 --splitk $sk --rotate 64 --reps 200 --no-stats; done
for sk in 1 2; do python tools/bench_one.py --hw 64 64 --cin 320 --cout 320 --k 1 --tile 3 --splitk $sk --rotate 64 --reps 200 --no-stats; done
