#!/bin/bash
# phase clocks of the workgroup kernel at N = 64 (two wavefronts write each counter: halve the times)
mkdir -p profiles/_ab gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DMPMPC_PHASE_CLOCK \
    -Iinclude -o profiles/_ab/P.so multi-purpose-mpc_amd/csrc/mpmpc_hip.hip || exit 1
MPMPC_PHASES_N=64 python profiles/phases.py 2 1024 2>&1 | head -22
