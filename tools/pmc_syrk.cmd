R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_syrk; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/time_syrk_only.py 1250000 1000 > $O/fetch.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/hit -- python3 $R/tools/time_syrk_only.py 1250000 1000 > $O/hit.log 2>&1 || exit 4
python3 $R/tools/pmc_summary.py $(find $O/fetch -name "*counter_collection.csv") k_syrk_lds
python3 $R/tools/pmc_summary.py $(find $O/hit -name "*counter_collection.csv") k_syrk_lds
