#!/bin/bash
# GPU call r05o: 512 members and more — band rows below their creator's threshold skipped (against the previous build), the phases
# of an iteration at 1024 members, the wide-member parity cases
O=gpurun_out/r05o; mkdir -p $O
B=profiles/ab/libswirld_hip_base.so
SWEEP_LIB=$B timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - > $O/ab_1024x2M.log 2>&1
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_TALLY_K=16,SW_SKIP=10 SW_TALLY_K=12,SW_SKIP=8 >> $O/ab_1024x2M.log 2>&1; cat $O/ab_1024x2M.log
SWEEP_LIB=$B timeout 300 python profiles/knob_sweep.py 700 1000000 3 -- - > $O/ab_700x1M.log 2>&1
timeout 300 python profiles/knob_sweep.py 700 1000000 3 -- - >> $O/ab_700x1M.log 2>&1; cat $O/ab_700x1M.log
SW_DEBUG_CLOCKS=2 SW_PIPE=1 timeout 200 python profiles/loop_phases.py 1024 2000000 > $O/loop_phases_1024.txt 2>&1; sed -n 1,12p $O/loop_phases_1024.txt; tail -8 $O/loop_phases_1024.txt
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_partition.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
