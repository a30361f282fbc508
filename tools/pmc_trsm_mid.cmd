# usage (GPU box): bash tools/pmc_trsm_mid.cmd  -> SQ counters of the one-pass solve chain at D = 384 (NB = 24) and 256 (NB = 16)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for D in 384 256; do
  O=$R/gpurun_out/pmc_trsm_$D; mkdir -p $O
  timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 $R/tools/check_trsm.py $D > $O/sq.log 2>&1 || exit 3
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/check_trsm.py $D > $O/kt.log 2>&1 || exit 4
  echo "== D=$D"
  python3 $R/tools/pmc_summary.py $(find $O/sq -name "*counter_collection.csv") k_trsm_rows_mid
  grep k_trsm_rows_mid $(find $O/kt -name "*kernel_stats.csv") | cut -c1-160
done
