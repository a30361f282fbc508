# usage (GPU box): bash tools/pmc_hess.cmd "base ks"   -> SQ counters of k_glm_hess_tile per dev build (vittles_amd/abl/lib_<name>.so)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for a in $1; do
  O=$R/gpurun_out/pmc_hess_$a; mkdir -p $O
  VITTLES_HIP_LIB=$R/vittles_amd/abl/lib_$a.so timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 $R/tools/time_hess_only.py 1000000 100 > $O/sq.log 2>&1 || exit 3
  VITTLES_HIP_LIB=$R/vittles_amd/abl/lib_$a.so timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/tools/time_hess_only.py 1000000 100 > $O/sq2.log 2>&1 || exit 4
  echo "== $a"
  python3 $R/tools/pmc_summary.py $(find $O/sq -name "*counter_collection.csv") k_glm_hess_tile
  python3 $R/tools/pmc_summary.py $(find $O/sq2 -name "*counter_collection.csv") k_glm_hess_tile
done
