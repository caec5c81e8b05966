# PMC passes over the 3-D scalar sweep: bash scripts/pmc_3d.sh <name> [bench args]  -> gpurun_out/<name>/pmc_summary.json
export TMPDIR=/tmp
N=${1:-pmc3d}; shift || true
O=$PWD/gpurun_out/$N; mkdir -p $O
BENCH="python3 bench.py --workload sphere3d_4.6M --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline $*"
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  t=$(echo $c | tr ' ' '_' | cut -c1-60)
  echo "$(date +%T) pmc $c" | tee -a $O/progress.log
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$t -o pmc -- $BENCH > $O/pmc_$t.log 2>&1 || echo "pass $c failed" | tee -a $O/progress.log
done
python3 scripts/summarize_pmc.py $O > $O/pmc_summary.json
rm -rf $O/pmc_*/ $O/pmc_*.log
echo done $N
