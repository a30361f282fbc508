#!/bin/bash
# usage (GPU box): bash tools/ab_hess32.sh "base abl1 ..." [D]  -> f32 Hessian / TRSM time per dev build
for a in $1; do
  VITTLES_HIP_LIB=$GRAFT_REPO_ROOT/vittles_amd/abl/lib_$a.so timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/time_hess32.py ${2:-100} 2>&1 | grep -v amdgpu.ids || exit 1
done
