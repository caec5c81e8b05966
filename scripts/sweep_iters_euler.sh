# blocks-per-wave scan of the single-kernel Euler sweep (run on the GPU box)
for it in 1 2 3 4 6; do
  IBH_SWEEP_ITERS=$it python bench.py --steps 200 --warmup 20 --no-cpu-baseline --residual euler ${1:+--workload $1} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('euler iters $it sweep_us', round(d['ms_per_step']*1e3,2), 'kernel_us', d['roofline']['kernel_us'])"
done
