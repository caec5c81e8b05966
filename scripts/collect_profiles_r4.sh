# Profile set of round 4 (run from the repo root on the GPU box):  bash scripts/collect_profiles_r4.sh <name> [quick|big|all]
# Per workload: the bench line, a rocprofv3 kernel trace (kernel_stats_<tag>.csv) and PMC passes in separate runs
# (pmc_summary_<tag>.json: FETCH_SIZE, WRITE_SIZE, SQ groups).  Writes gpurun_out/<name>/...; copy what is to be kept into
# profiles/<name>/.  Every step appends to progress.log.
export TMPDIR=/tmp
N=${1:-r4}; O=$PWD/gpurun_out/$N; mkdir -p $O
say() { echo "$(date +%T) $*" | tee -a $O/progress.log; }
GROUPS_PMC=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
            "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE")
kernel_trace() {   # tag, bench args...
  local tag=$1; shift
  say "kernel trace $tag"
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o kt -- python3 bench.py --no-cpu-baseline $* > $O/kt_$tag.log 2>&1 || say "kernel trace failed: $tag"
  cp $(find $O/kt_$tag -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$tag.csv 2>/dev/null
  rm -rf $O/kt_$tag $O/kt_$tag.log
}
pmc_passes() {     # tag, n_groups, bench args...
  local tag=$1 ng=$2; shift 2
  mkdir -p $O/p_$tag
  local i=0
  for c in "${GROUPS_PMC[@]}"; do
    [ $i -ge $ng ] && break; i=$((i+1))
    local t=$(echo $c | tr ' ' '_' | cut -c1-60)
    say "pmc $tag $c"
    timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d $O/p_$tag/pmc_$t -o pmc -- python3 bench.py --no-cpu-baseline $* > $O/p_$tag/pmc_$t.log 2>&1 || say "pass failed: $tag $c"
  done
  python3 scripts/summarize_pmc.py $O/p_$tag > $O/pmc_summary_$tag.json
  rm -rf $O/p_$tag
}
profile_one() {   # tag, bench args...
  local tag=$1; shift
  say "bench $tag"; python3 bench.py --no-cpu-baseline $* > $O/bench_$tag.json 2>>$O/bench.err
  kernel_trace $tag $* --repeats 3
  pmc_passes $tag 6 $* --repeats 3
}
MODE=${2:-all}
if [ "$MODE" = "quick" ] || [ "$MODE" = "all" ]; then
say "headline with the cpu baseline"; python3 bench.py > $O/bench.json 2>$O/bench.err
profile_one headline
profile_one euler2d --residual euler
profile_one 3d_euler_4.6M --workload sphere3d_4.6M --residual euler --steps 50 --warmup 5
profile_one 3d_4.6M --workload sphere3d_4.6M --steps 100 --warmup 10
profile_one 3.47M --workload rae2822_3.47M
say "march"; python3 bench.py --no-cpu-baseline --step march > $O/bench_march.json 2>>$O/bench.err
python3 bench.py --no-cpu-baseline --step march --dt-every 10 > $O/bench_march_dt_every_10.json 2>>$O/bench.err
python3 bench.py --no-cpu-baseline --step march --dt-separate > $O/bench_march_dt_separate_launches.json 2>>$O/bench.err
kernel_trace march --step march --repeats 3
say probes
python3 scripts/probe_3d_euler.py sphere3d_1.6M > $O/probe_3d_euler_1.6M.json 2>>$O/bench.err
python3 scripts/probe_3d_euler.py sphere3d_4.6M > $O/probe_3d_euler_4.6M.json 2>>$O/bench.err
python3 scripts/wave_timeline_3d.py > $O/wave_timeline_3d_euler_4.6M.json 2>>$O/bench.err
fi
if [ "$MODE" = "big" ] || [ "$MODE" = "all" ]; then
  say "3d euler 33M"; bash scripts/with_heartbeat.sh $O/progress.log python3 bench.py --no-cpu-baseline --workload sphere3d_33M --residual euler --steps 20 --warmup 3 --repeats 5 > $O/bench_3d_euler_33M.json 2>>$O/bench.err
  pmc_passes 3d_euler_33M 2 --workload sphere3d_33M --residual euler --steps 10 --warmup 2 --repeats 2
  say config4; python3 bench.py --no-cpu-baseline --workload sphere3d_8M --residual euler --step config4 --steps 20 --warmup 3 --repeats 5 > $O/bench_config4_8M.json 2>>$O/bench.err
  kernel_trace config4_8M --workload sphere3d_8M --residual euler --step config4 --steps 20 --warmup 3 --repeats 1
  say "config5 8M"; bash scripts/with_heartbeat.sh $O/progress.log python3 bench.py --no-cpu-baseline --workload sphere3d_8M --residual euler --step config5 --steps 3 --warmup 1 --repeats 2 > $O/bench_config5_8M.json 2>>$O/bench.err
  kernel_trace config5_8M --workload sphere3d_8M --residual euler --step config5 --steps 3 --warmup 1 --repeats 1
fi
if [ "$MODE" = "c5" ]; then
  say "config5 33M"; bash scripts/with_heartbeat.sh $O/progress.log python3 bench.py --no-cpu-baseline --workload sphere3d_33M --residual euler --step config5 --steps 3 --warmup 1 --repeats 2 > $O/bench_config5_33M.json 2>>$O/bench.err
fi
if [ "$MODE" = "c5kt" ]; then
  # (the 33.6 M-cell domain takes minutes to build and rocprofv3 must start python itself: a background writer keeps the
  # run from looking hung)
  ( while sleep 60; do echo "$(date +%T) still tracing config5_33M" >> $O/progress.log; done ) &
  HB=$!
  trap "kill $HB 2>/dev/null" EXIT
  kernel_trace config5_33M --workload sphere3d_33M --residual euler --step config5 --steps 3 --warmup 1 --repeats 1
fi
if [ "$MODE" = "28M" ]; then
  say 28M; bash scripts/with_heartbeat.sh $O/progress.log python3 bench.py --no-cpu-baseline --workload rae2822_28M --steps 50 --warmup 5 --repeats 5 > $O/bench_28M.json 2>>$O/bench.err
  say "3d 33M"; bash scripts/with_heartbeat.sh $O/progress.log python3 bench.py --no-cpu-baseline --workload sphere3d_33M --steps 50 --warmup 5 --repeats 5 > $O/bench_3d_33M.json 2>>$O/bench.err
fi
say done
