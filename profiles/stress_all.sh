#!/bin/bash
# the device-against-emulation sweep over several seeds -> profiles/<round>/stress.txt (via gpurun_out/stress.txt)
mkdir -p gpurun_out
python -c "import sys; sys.path.insert(0, 'multi-purpose-mpc_amd'); import mpmpc; print('#', mpmpc.load_library().mpmpc_version().decode())" > gpurun_out/stress.txt
for s in $(seq 0 ${1:-19}); do timeout 600 python profiles/stress.py $s 2>&1 | tail -2 >> gpurun_out/stress.txt; done
tail -6 gpurun_out/stress.txt
