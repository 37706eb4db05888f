#!/bin/bash
# Build an A/B variant of libpir_hip.so with extra compiler flags:  tools/build_variant.sh NAME -DFLAG [-DFLAG ...]
# -> variants/NAME/libpir_hip.so (load it with PIR_HIP_LIB=variants/NAME/libpir_hip.so).  variants/ is not committed.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=variants/$name
mkdir -p "$out"
pids=()
for f in sdf_eval trace ggx sdf_backward wgrad visibility mlp shade_inputs embed edge_walk prof; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment "$@" -c pir_amd/csrc/$f.hip -o "$out/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libpir_hip.so" "$out"/*.o
echo "built $out/libpir_hip.so"
