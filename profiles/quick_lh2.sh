#!/bin/bash
timeout 1500 python -m pytest tests/test_long_horizon.py -m gpu -q 2>&1 | tail -6
timeout 600 python profiles/long_horizon_timing.py 2>&1 | grep "lanes" | grep "B  8192"
