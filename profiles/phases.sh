#!/bin/bash
# per-wave phase clocks of K2 (profiling build of the library, -DMPMPC_PHASE_CLOCK) for the configurations of a round:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash profiles/phases.sh r2'
R=${1:-r6}
mkdir -p profiles/_ab gpurun_out/$R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DMPMPC_PHASE_CLOCK \
    -Iinclude -o profiles/_ab/P.so multi-purpose-mpc_amd/csrc/mpmpc_hip.hip || exit 1
python profiles/phases.py 2 > gpurun_out/$R/phases_2.txt
MPMPC_PHASES_PIPELINE=3 python profiles/phases.py 2 > gpurun_out/$R/phases_2_packed.txt
python profiles/phases.py 3 > gpurun_out/$R/phases_3.txt
python profiles/phases.py 4 > gpurun_out/$R/phases_4.txt
python profiles/phases.py 2 65536 > gpurun_out/$R/phases_2_65536.txt
head -22 gpurun_out/$R/phases_2.txt; head -12 gpurun_out/$R/phases_4.txt
