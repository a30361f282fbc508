# usage: bash tools/build_abl.cmd "0 1 3 ..."   -> vittles_amd/abl/lib_<mask>.so (dev-only ablation builds of vt_hessian.hip)
set -e
cd /root/repo/vittles_amd/csrc
mkdir -p ../abl
OTHERS=$(ls *.o | grep -v vt_hessian.o)
for m in $1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DVT_ABL=$m -c vt_hessian.hip -o /tmp/vt_hessian_abl$m.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_$m.so /tmp/vt_hessian_abl$m.o $OTHERS
done
ls -la ../abl
