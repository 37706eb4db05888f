#!/bin/bash
# development aid: build variants/<name>.so with extra -D flags:  tools/build_variant.sh name -DPIR_ABL_NOSTORE ...
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p variants/obj_$name
for f in sdf_eval trace ggx sdf_backward wgrad visibility mlp prof; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment "$@" -c pir_amd/csrc/$f.hip -o variants/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so variants/obj_$name/*.o
rm -rf variants/obj_$name
echo built variants/$name.so
