#!/bin/bash
# usage: bash tools/build_dev32.sh <name> [extra -D flags...]  -> vittles_amd/abl/lib_<name>.so
# dev build of vt_f32.hip with only one block count of the f32 Hessian compiled (DEVNB, default 7: D = 97..112)
set -e
cd /root/repo/vittles_amd/csrc
mkdir -p ../abl
name=$1; shift
OTHERS=$(ls *.o | grep -v vt_f32.o)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DVT_F32_DEV_NB=${DEVNB:-7} "$@" -c vt_f32.hip -o /tmp/vt_f32_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_$name.so /tmp/vt_f32_$name.o $OTHERS
ls -la ../abl/lib_$name.so
