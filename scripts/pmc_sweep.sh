# PMC passes over the headline bench (quad sweep kernel): bash scripts/pmc_sweep.sh <name> [bench args]
#   -> gpurun_out/<name>/pmc_summary.json (per-launch averages) ; counters in separate passes as the guide prescribes
set -e
export TMPDIR=/tmp
N=${1:-pmc}; shift || true
O=$PWD/gpurun_out/$N; mkdir -p $O
BENCH="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline $*"
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU"; do
  t=$(echo $c | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$t -o pmc -- $BENCH > $O/pmc_$t.log 2>&1 || echo "pass $c failed"
done
python3 scripts/summarize_pmc.py $O > $O/pmc_summary.json
rm -rf $O/pmc_*/
echo done $N
