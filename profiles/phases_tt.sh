R=$1
mkdir -p profiles/_ab gpurun_out/$R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DMPMPC_PHASE_CLOCK \
    -Iinclude -o profiles/_ab/P.so multi-purpose-mpc_amd/csrc/mpmpc_hip.hip || exit 1
python profiles/phases.py 3 1024 > gpurun_out/$R/phases_3_1024.txt
python profiles/phases.py 3 > gpurun_out/$R/phases_3.txt
cat gpurun_out/$R/phases_3_1024.txt; head -22 gpurun_out/$R/phases_3.txt
