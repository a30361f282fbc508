# usage (GPU box): bash tools/pmc_trsm_large.cmd  -> HBM traffic of the blocked solve at N = 1.25e6, D = 1000 (FETCH_SIZE / WRITE_SIZE, separate passes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_trsm_large; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/check_trsm.py 1000 1250000 > $O/fetch.log 2>&1 || exit 3
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/check_trsm.py 1000 1250000 > $O/write.log 2>&1 || exit 4
python3 $R/tools/pmc_summary.py $(find $O/fetch -name "*counter_collection.csv") k_
python3 $R/tools/pmc_summary.py $(find $O/write -name "*counter_collection.csv") k_
