#!/bin/bash
# usage (on the GPU box): bash tools/prof_hess.cmd "0 1 3"  -> kernel durations of the fused Hessian per ablation build
cd /tmp && export TMPDIR=/tmp
for a in $1; do
  VITTLES_HIP_LIB=$GRAFT_REPO_ROOT/vittles_amd/abl/lib_$a.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hp$a -- python3 $GRAFT_REPO_ROOT/tools/time_hess.py > /dev/null 2>&1 || exit 1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/hp$a -name "*kernel_stats.csv" | head -1)
  echo "ablate $a"; grep -E "Name|k_glm_hess|k_reduce|k_hess_fin" $f | python3 -c "import csv,sys; [print(\"  \", r[\"Name\"][27:70], round(float(r[\"AverageNs\"])/1e3,1), \"us\") for r in csv.DictReader(sys.stdin)]"
done
