#!/bin/bash
# GPU call r05s: k_tally_search (SW_TALLY_IMPL=3: classification by popcount bounds + bisection) — parity, then 1024 / 700 / 256 members
O=gpurun_out/r05s; mkdir -p $O
SW_TALLY_IMPL=3 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity_impl3.log 2>&1; tail -3 $O/pytest_parity_impl3.log
timeout 600 python -m pytest tests/test_gpu_random.py -m gpu -x -q > $O/pytest_random.log 2>&1; tail -3 $O/pytest_random.log
timeout 500 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_TALLY_IMPL=3 SW_TALLY_IMPL=3,SW_TALLY_K=16,SW_SKIP=8 SW_TALLY_IMPL=3,SW_TALLY_K=20,SW_SKIP=6 SW_TALLY_IMPL=3,SW_TALLY_K=12,SW_SKIP=8 - > $O/knobs_1024x2M.log 2>&1; cat $O/knobs_1024x2M.log
timeout 300 python profiles/knob_sweep.py 700 1000000 3 -- - SW_TALLY_IMPL=3 > $O/knobs_700x1M.log 2>&1; cat $O/knobs_700x1M.log
timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- - SW_TALLY_IMPL=3 SW_TALLY_IMPL=3,SW_TALLY_K=32 > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
SW_TALLY_IMPL=3 timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -k "1024 or coin" > $O/pytest_wide_impl3.log 2>&1; tail -3 $O/pytest_wide_impl3.log
