# blocks-per-wave scan of the single-kernel sweep (run on the GPU box): bash scripts/sweep_iters.sh [workload]
W=${1:-rae2822_0.87M}
for it in 1 2 3 4 6 8; do
  IBH_SWEEP_ITERS=$it python bench.py --steps 400 --warmup 40 --no-cpu-baseline --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W iters $it sweep_us', round(d['ms_per_step']*1e3,2), 'kernel_us', d['roofline']['kernel_us'])"
done
