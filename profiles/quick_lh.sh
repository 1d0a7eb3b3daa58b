#!/bin/bash
# same-box check of the long-horizon kernels: device tests, then launch times around the 63 / 64 boundary
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python profiles/long_horizon_timing.py 2>&1 | tail -24
