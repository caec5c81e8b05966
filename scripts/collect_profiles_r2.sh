# Profile set of round 2 (run from the repo root on the GPU box):  bash scripts/collect_profiles_r2.sh <name>
# Writes gpurun_out/<name>/...; copy what is to be kept into profiles/<name>/.  Every step appends to progress.log.
set -e
export TMPDIR=/tmp
N=${1:-r2}; O=$PWD/gpurun_out/$N; mkdir -p $O
say() { echo "$(date +%T) $*" | tee -a $O/progress.log; }
B="python3 bench.py --no-cpu-baseline"
say bench; python3 bench.py > $O/bench.json 2>$O/bench.err
say 3.47M; $B --workload rae2822_3.47M > $O/bench_3.47M.json 2>>$O/bench.err
say euler; $B --residual euler > $O/bench_euler.json 2>>$O/bench.err
say 3d; $B --workload sphere3d_4.6M > $O/bench_3d_4.6M.json 2>>$O/bench.err
say 3d euler; $B --steps 100 --warmup 10 --residual euler --workload sphere3d_1.6M > $O/bench_3d_euler_1.6M.json 2>>$O/bench.err
say config4; $B --workload sphere3d_8M --residual euler --step config4 --steps 20 --warmup 3 --repeats 5 > $O/bench_config4_8M.json 2>>$O/bench.err
say config5; $B --workload sphere3d_4.6M --residual euler --step config5 --steps 5 --warmup 1 --repeats 3 > $O/bench_config5_4.6M.json 2>>$O/bench.err
say config5 33M; bash scripts/with_heartbeat.sh $O/progress.log $B --workload sphere3d_33M --residual euler --step config5 --steps 3 --warmup 1 --repeats 2 > $O/bench_config5_33M.json 2>>$O/bench.err
say 3d 33M; bash scripts/with_heartbeat.sh $O/progress.log $B --workload sphere3d_33M --steps 50 --warmup 5 --repeats 5 > $O/bench_3d_33M.json 2>>$O/bench.err
say 3d euler 33M; bash scripts/with_heartbeat.sh $O/progress.log $B --workload sphere3d_33M --residual euler --steps 20 --warmup 3 --repeats 5 > $O/bench_3d_euler_33M.json 2>>$O/bench.err
say closure; python3 scripts/probe_closure.py > $O/probe_closure.json 2>>$O/bench.err
say 28M; $B --workload rae2822_28M --steps 50 --warmup 5 --repeats 5 > $O/bench_28M.json 2>>$O/bench.err
say probes 3d / euler; python3 scripts/probe_3d.py > $O/probe_3d_4.6M.json 2>>$O/bench.err
python3 scripts/probe_euler.py > $O/probe_euler.json 2>>$O/bench.err
say probe; python3 scripts/probe_sweep.py > $O/probe_0.87M.json 2>>$O/bench.err
QUAD_TUNE=1 python3 scripts/probe_sweep.py > $O/probe_parts_0.87M.json 2>>$O/bench.err
python3 scripts/probe_sweep.py rae2822_3.47M > $O/probe_3.47M.json 2>>$O/bench.err
say timeline; python3 scripts/wave_timeline.py > $O/wave_timeline_0.87M.json 2>>$O/bench.err
say kernel-trace; rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline > $O/kt.log 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  t=$(echo $c | tr ' ' '_' | cut -c1-60)
  say "pmc $c"
  timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$t -o pmc -- python3 bench.py --no-cpu-baseline --repeats 3 > $O/pmc_$t.log 2>&1 || say "pass failed: $c"
done
python3 scripts/summarize_pmc.py $O > $O/pmc_summary.json
rm -rf $O/kt $O/pmc_*/ $O/pmc_*.log
say done
