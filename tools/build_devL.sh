#!/bin/bash
# usage: bash tools/build_devL.sh <name> [extra -D flags...]  -> vittles_amd/abl/lib_<name>.so  (dev build of vt_large.hip)
set -e
cd /root/repo/vittles_amd/csrc
mkdir -p ../abl
name=$1; shift
OTHERS=$(ls *.o | grep -v vt_large.o)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c vt_large.hip -o /tmp/vt_large_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_$name.so /tmp/vt_large_$name.o $OTHERS
ls -la ../abl/lib_$name.so
