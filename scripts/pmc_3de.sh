# PMC passes over the 3-D Euler column sweep for one variant: bash scripts/pmc_3de.sh <name> <quad_variant> [workload]
export TMPDIR=/tmp
N=$1; V=$2; W=${3:-sphere3d_4.6M}
export IBH_QUAD_VARIANT=$V
O=$PWD/gpurun_out/$N; mkdir -p $O
BENCH="python3 bench.py --workload $W --residual euler --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline"
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  t=$(echo $c | tr ' ' '_' | cut -c1-60)
  echo "$(date +%T) pmc $c" | tee -a $O/progress.log
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$t -o pmc -- $BENCH > $O/pmc_$t.log 2>&1 || echo "pass $c failed" | tee -a $O/progress.log
done
python3 scripts/summarize_pmc.py $O > $O/pmc_summary.json
rm -rf $O/pmc_*/ $O/pmc_*.log
echo done $N
