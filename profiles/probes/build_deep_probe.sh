#!/bin/bash
# builds the deep-kernel probe variants into profiles/probes/ (run here: hipcc cross-compiles; the binaries travel with gpurun)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-result"
hipcc $F deep_probe.hip -o deep_probe &
hipcc $F -DDEEP_STAMPS deep_probe.hip -o deep_probe_stamps &
hipcc $F -DDEEP_STAMPS -DDEEP_NO_A deep_probe.hip -o deep_probe_stamps_noa &
hipcc $F -DDEEP_STAMPS -DDEEP_NO_B deep_probe.hip -o deep_probe_stamps_nob &
wait
hipcc $F -DDEEP_STAMPS -DDEEP_NO_MFMA deep_probe.hip -o deep_probe_stamps_nomfma &
hipcc $F -DDEEP_STAMPS -DDEEP_A1 deep_probe.hip -o deep_probe_stamps_a1 &
wait
ls deep_probe*
