#!/bin/bash
# libnunif_hip_trace.so: the product library with depth_mlp.hip compiled -DNUNIF_MLP_TRACE (phase timestamps of the split MLP
# kernel printed to stderr).  Run from anywhere; pick it with NUNIF_HIP_LIB=<repo>/nunif_amd/libnunif_hip_trace.so.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
python -m nunif_amd.build > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -Wno-unused-function -fno-gpu-rdc -DNUNIF_MLP_TRACE \
    -I "$REPO/nunif_amd/csrc" -c "$REPO/nunif_amd/csrc/depth_mlp.hip" -o /tmp/depth_mlp_trace.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$REPO/nunif_amd/libnunif_hip_trace.so" \
    $(ls "$REPO"/nunif_amd/csrc/_obj/*.o | grep -v depth_mlp) /tmp/depth_mlp_trace.o
echo "built $REPO/nunif_amd/libnunif_hip_trace.so"
