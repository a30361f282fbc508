#!/bin/bash
# usage: bash tools/build_devH.sh <name> [extra -D flags...]  -> vittles_amd/abl/lib_<name>.so  (dev build of vt_hessian.hip;
# add -DVT_HESS_DEV_NB=7 to compile one tile-kernel block count only)
set -e
cd /root/repo/vittles_amd/csrc
mkdir -p ../abl
name=$1; shift
OTHERS=$(ls *.o | grep -v vt_hessian.o)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c vt_hessian.hip -o /tmp/vt_hessian_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_$name.so /tmp/vt_hessian_$name.o $OTHERS
ls -la ../abl/lib_$name.so
