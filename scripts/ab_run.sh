# same-box A/B of library variants: bash scripts/ab_run.sh "<bench args>" name1 name2 ...  (3 alternations)
A="$1"; shift
for i in 1 2 3; do for n in "$@"; do
  IBHIP_LIB=$PWD/immersedboundary.jl_amd/lib_$n.so python bench.py --steps 400 --warmup 40 --no-cpu-baseline $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', 'sweep_us', round(d['ms_per_step']*1e3,2), 'kernel_us', d['roofline']['kernel_us'])"
done; done
