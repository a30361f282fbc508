#!/bin/bash
# usage (GPU box): bash tools/ab_hess.sh "base abl4 ..." [D]  -> whole-call time of the fused Hessian per dev build
for a in $1; do
  VITTLES_HIP_LIB=$GRAFT_REPO_ROOT/vittles_amd/abl/lib_$a.so timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/time_hess.py ${2:-100} 2>/dev/null || exit 1
done
