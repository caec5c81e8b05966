# PMC passes of counter groups over any script: PMC_KERNEL_RE='k_viscous\w*' bash scripts/pmc_script.sh <name> "<G1>;<G2>" script.py args...
export TMPDIR=/tmp
N=$1; GROUPS_=$2; shift 2
O=$PWD/gpurun_out/$N; mkdir -p $O
IFS=';' read -ra GS <<< "$GROUPS_"
for c in "${GS[@]}"; do
  t=$(echo $c | tr ' ' '_' | cut -c1-60)
  echo "$(date +%T) pmc $c" | tee -a $O/progress.log
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$t -o pmc -- python3 $* > $O/pmc_$t.log 2>&1 || { echo "pass $c failed" | tee -a $O/progress.log; tail -3 $O/pmc_$t.log >> $O/progress.log; }
done
python3 scripts/summarize_pmc.py $O > $O/pmc_summary.json
rm -rf $O/pmc_*/ $O/pmc_*.log
echo done $N
