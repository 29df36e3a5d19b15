#!/bin/bash
# GPU call r05e: popcount bounds in front of the mask gathers of the one-wave-per-slot tally (SW_TALLY_FILTER=1) against the
# two-level search, windows of 28 ... 60 slots
O=gpurun_out/r05e; mkdir -p $O
timeout 400 python profiles/knob_sweep.py 256 1000000 9 -- - SW_TALLY_IMPL=1,SW_TALLY_K=28 SW_TALLY_IMPL=1,SW_TALLY_K=28,SW_TALLY_FILTER=1 SW_TALLY_IMPL=1,SW_TALLY_K=32,SW_TALLY_FILTER=1 SW_TALLY_IMPL=1,SW_TALLY_K=24,SW_TALLY_FILTER=1 SW_TALLY_IMPL=1,SW_TALLY_K=40,SW_TALLY_FILTER=1 SW_TALLY_IMPL=1,SW_TALLY_K=60,SW_TALLY_FILTER=1 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
SW_TALLY_IMPL=1 SW_TALLY_K=32 SW_TALLY_FILTER=1 timeout 100 python profiles/loop_phases.py > $O/loop_phases_filter.txt 2>&1; sed -n 1,40p $O/loop_phases_filter.txt
timeout 300 python profiles/knob_sweep.py 64 100000 9 -- - SW_TALLY_FILTER=1 SW_TALLY_FILTER=1,SW_TALLY_K=40 - > $O/knobs_64x100k.log 2>&1; cat $O/knobs_64x100k.log
timeout 400 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_TALLY_FILTER=1 SW_TALLY_FILTER=1,SW_TALLY_K=16,SW_SKIP=8 SW_TALLY_FILTER=1,SW_TALLY_K=24,SW_SKIP=4 SW_TALLY_FILTER=1,SW_TALLY_K=28,SW_SKIP=1 > $O/knobs_1024x2M.log 2>&1; cat $O/knobs_1024x2M.log
GEN_MODE=1 GEN_P0=0.02 timeout 200 python profiles/knob_sweep.py 256 1000000 5 -- - SW_TALLY_IMPL=1,SW_TALLY_K=32,SW_TALLY_FILTER=1 SW_TALLY_IMPL=1,SW_TALLY_K=60,SW_TALLY_FILTER=1 > $O/knobs_cliques_256x1M.log 2>&1; cat $O/knobs_cliques_256x1M.log
GEN_MODE=2 GEN_P0=0.35 GEN_P1=0.02 timeout 200 python profiles/knob_sweep.py 256 1000000 5 -- - SW_TALLY_FILTER=1 SW_TALLY_FILTER=1,SW_TALLY_K=40 > $O/knobs_coin_256x1M.log 2>&1; cat $O/knobs_coin_256x1M.log
SW_TALLY_FILTER=1 SW_TALLY_IMPL=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
