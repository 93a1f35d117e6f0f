# Four --pmc passes over the concurrent step (counters only, no other trace domain)
# -> gpurun_out/pmc/report.txt (profiles/r05_pmc_concurrent_step.txt)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/pmc; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -- python tools/time_train_step.py concurrent graph > $O/p$i.log 2>&1
  f=$(ls $O/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f step > $O/summary$i.csv
  rm -rf $O/p$i
done
python tools/pmc_step_report.py $O/summary*.csv > $O/report.txt
cat $O/report.txt
