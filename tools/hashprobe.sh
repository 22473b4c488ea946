S="--workload avatar --grad-hash --gaussians 20000 --steps 2 --warmup 1 --no-cpu-baseline"
g() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], j['grad_sha256'][:16])" "$1"; }
python bench.py $S 2>/dev/null | g plain1
python bench.py $S 2>/dev/null | g plain2
python bench.py $S --views-per-step 1 --streams 1 2>/dev/null | g k1_a
python bench.py $S --views-per-step 1 --streams 1 2>/dev/null | g k1_b
SINGS_BENCH_FORCE_DIST=1 python bench.py $S --reduce-chunks 1 2>/dev/null | g force_chunks1
SINGS_BENCH_FORCE_DIST=1 python bench.py $S 2>/dev/null | g force_chunks4
SINGS_BENCH_FORCE_DIST=1 python bench.py $S --views-per-step 1 --streams 1 2>/dev/null | g force_k1
