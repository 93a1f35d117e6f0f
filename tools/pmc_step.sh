# Counter passes (--pmc only, no other trace domain) over one fused training step:
#   bash tools/pmc_step.sh <concurrent|autoregressive|LSTM> [outdir]
# -> <outdir>/report.txt  (profiles/r06_pmc_ar_step.txt is the autoregressive one)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
MODE=${1:-autoregressive}
O=${2:-gpurun_out/pmc_$MODE}; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -- python tools/time_train_step.py $MODE graph > $O/p$i.log 2>&1
  f=$(ls $O/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f step > $O/summary$i.csv || tail -5 $O/p$i.log
  rm -rf $O/p$i
done
python tools/pmc_step_report.py $MODE $O/summary*.csv > $O/report.txt
cat $O/report.txt
