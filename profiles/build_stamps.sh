#!/bin/bash
# Experiment build of the library with -DVNET_STAMPS (s_memtime stamps inside the persistent bf16 kernels) into
# profiles/probes/libvnet_hip_stamps.so; use with VNET_HIP_LIB=$PWD/profiles/probes/libvnet_hip_stamps.so python profiles/step_stamps.py
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
for f in conv_mfma conv_x3 conv_b16 conv2_b16 elementwise input_block; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result ${DEFS:--DVNET_STAMPS} $EXTRA \
      -Iinclude -c vnet_tensorflow_amd/csrc/$f.hip -o $T/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $T/*.o -o profiles/probes/${OUT:-libvnet_hip_stamps.so}
rm -rf $T
ls -la profiles/probes/${OUT:-libvnet_hip_stamps.so}
