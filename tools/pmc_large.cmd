R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_large; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 $R/tools/time_large.py 1250000 1000 > $O/sq.log 2>&1 || exit 3
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $O/sq2 -- python3 $R/tools/time_large.py 1250000 1000 > $O/sq2.log 2>&1 || exit 4
python3 $R/tools/pmc_summary.py $(find $O/sq -name "*counter_collection.csv") k_syrk_lds
python3 $R/tools/pmc_summary.py $(find $O/sq -name "*counter_collection.csv") k_gemm_nt_lds
python3 $R/tools/pmc_summary.py $(find $O/sq -name "*counter_collection.csv") k_trsm_rows_mid
python3 $R/tools/pmc_summary.py $(find $O/sq2 -name "*counter_collection.csv") k_syrk_lds
python3 $R/tools/pmc_summary.py $(find $O/sq2 -name "*counter_collection.csv") k_gemm_nt_lds
