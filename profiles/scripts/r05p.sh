#!/bin/bash
# GPU call r05p: wide-member parity with the band skip (the device was not found by the test process of r05o: once more, alone)
O=gpurun_out/r05p; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_partition.py tests/test_gpu_parity.py -m gpu -x -q -s > $O/pytest.log 2>&1; tail -5 $O/pytest.log
dmesg 2>/dev/null | tail -5
