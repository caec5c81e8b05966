# Collect one profile set on the GPU box (run from the repo root):
#   bash scripts/collect_profiles.sh r1_v3
# Writes gpurun_out/<name>/{kernel_stats.csv,pmc_summary.json,bench*.json}; copy what is to be kept into profiles/<name>/.
set -e
export TMPDIR=/tmp
N=${1:-set}; O=$PWD/gpurun_out/$N; mkdir -p $O
BENCH="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline"
if [ -n "$PMC_ONLY" ]; then SKIP=1; fi
if [ -z "$SKIP" ]; then
python3 bench.py --steps 400 --warmup 40 > $O/bench.json 2>$O/bench.err
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload rae2822_3.47M > $O/bench_3.47M.json 2>>$O/bench.err
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --residual euler > $O/bench_euler.json 2>>$O/bench.err
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload sphere3d_1.6M > $O/bench_3d_1.6M.json 2>>$O/bench.err
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload sphere3d_4.6M > $O/bench_3d_4.6M.json 2>>$O/bench.err
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --residual euler --workload rae2822_3.47M > $O/bench_euler_3.47M.json 2>>$O/bench.err
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --residual euler --workload sphere3d_1.6M > $O/bench_3d_euler_1.6M.json 2>>$O/bench.err
python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-fuse > $O/bench_two_kernel.json 2>>$O/bench.err
python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline --workload rae2822_37k > $O/bench_37k.json 2>>$O/bench.err
for n in 2 8; do python3 scripts/mixed_ab.py $n 2>/dev/null; done > $O/per_rank_sweeps.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kte -o kte -- $BENCH --residual euler > $O/kte.log 2>&1
cp $(find $O/kte -name '*kernel_stats.csv' | head -1) $O/kernel_stats_euler.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $BENCH > $O/kt.log 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
fi
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  t=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$t -o pmc -- $BENCH > $O/pmc_$t.log 2>&1
done
python3 scripts/summarize_pmc.py $O > $O/pmc_summary.json
rm -rf $O/kt $O/kte $O/pmc_*/ 
echo done $N
