#!/bin/bash
# GPU call r05q: band groups of 4 events beyond 256 members (against the previous build); wide-member parity (torch imported before the oracle pool)
O=gpurun_out/r05q; mkdir -p $O
B=profiles/ab/libswirld_hip_base.so
SWEEP_LIB=$B timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - > $O/ab_1024x2M.log 2>&1
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_BAND_BLOCKS=512 >> $O/ab_1024x2M.log 2>&1; cat $O/ab_1024x2M.log
SWEEP_LIB=$B timeout 300 python profiles/knob_sweep.py 700 1000000 3 -- - > $O/ab_700x1M.log 2>&1
timeout 300 python profiles/knob_sweep.py 700 1000000 3 -- - >> $O/ab_700x1M.log 2>&1; cat $O/ab_700x1M.log
timeout 1200 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_partition.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
