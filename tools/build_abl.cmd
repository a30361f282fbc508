# usage: bash tools/build_abl.cmd <file-stem> <macro> "0 1 3 ..."   -> vittles_amd/abl/lib_<mask>.so (dev-only ablation builds)
set -e
cd /root/repo/vittles_amd/csrc
mkdir -p ../abl
OTHERS=$(ls *.o | grep -v $1.o)
for m in $3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D$2=$m -c $1.hip -o /tmp/$1_abl$m.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_$m.so /tmp/$1_abl$m.o $OTHERS
done
ls ../abl
