#!/bin/bash
# GPU call r05u: k_tally_search with the classification of a wave's three slots in one batch of loads, <= 128 VGPRs
O=gpurun_out/r05u; mkdir -p $O
timeout 500 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_TALLY_IMPL=3 SW_TALLY_IMPL=3,SW_TALLY_K=16,SW_SKIP=8 - > $O/knobs_1024x2M.log 2>&1; cat $O/knobs_1024x2M.log
timeout 300 python profiles/knob_sweep.py 700 1000000 3 -- - SW_TALLY_IMPL=3 > $O/knobs_700x1M.log 2>&1; cat $O/knobs_700x1M.log
GEN_MODE=2 GEN_P0=0.40 GEN_P1=0.02 timeout 300 python profiles/knob_sweep.py 1024 2000000 2 -- - SW_TALLY_IMPL=3 > $O/knobs_coin_1024x2M.log 2>&1; cat $O/knobs_coin_1024x2M.log
SW_TALLY_IMPL=3 timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_partition.py -m gpu -x -q -k "1024 or coin or members or partition" > $O/pytest_wide_impl3.log 2>&1; tail -3 $O/pytest_wide_impl3.log
