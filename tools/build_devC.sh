#!/bin/bash
# usage: bash tools/build_devC.sh <name> [extra -D flags...]  -> vittles_amd/abl/lib_<name>.so  (dev build of vt_chol.hip)
set -e
cd /root/repo/vittles_amd/csrc
mkdir -p ../abl
name=$1; shift
OTHERS=$(ls *.o | grep -v vt_chol.o)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c vt_chol.hip -o /tmp/vt_chol_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_$name.so /tmp/vt_chol_$name.o $OTHERS
ls -la ../abl/lib_$name.so
