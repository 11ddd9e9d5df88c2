#!/bin/bash
# s_memtime stamps of four consecutive (item, chunk) steps of one workgroup of conv5_x3_kernel (csrc/conv_x3.h, -DVNET_STAMPS):
#   bash profiles/x3_stamps.sh            (on the GPU box; builds profiles/probes/libvnet_hip_x3stamps.so from the in-tree objects)
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -DVNET_STAMPS $EXTRA \
    -c vnet_tensorflow_amd/csrc/conv_x3.hip -o /tmp/conv_x3_stamps.o
cd vnet_tensorflow_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_mfma.o /tmp/conv_x3_stamps.o conv_b16.o conv2_b16.o elementwise.o input_block.o \
    -o ../../profiles/probes/libvnet_hip_x3stamps.so
cd ../..
VNET_HIP_LIB=$PWD/profiles/probes/libvnet_hip_x3stamps.so python profiles/x3_stamps.py "$@"
