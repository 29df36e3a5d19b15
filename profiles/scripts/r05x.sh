#!/bin/bash
# GPU call r05x: chained loops with a BRIDGE shot behind the chained start (a refusal no longer costs a whole shot): tests, then speed with exact and short predictions
O=gpurun_out/r05x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_errors.py -m gpu -x -q > $O/pytest_chain.log 2>&1; tail -3 $O/pytest_chain.log
timeout 400 python profiles/knob_sweep.py 256 1000000 11 -- - SW_CHAIN=1 SW_CHAIN=1,SW_SHOT_EXTRA=0 SW_CHAIN=1,SW_SHOT_EXTRA=1 SW_CHAIN=0,SW_SHOT_EXTRA=0 SW_CHAIN=1,SW_BRIDGE=8 SW_CHAIN=1,SW_BRIDGE=32 SW_CHAIN=1,SW_SHOT_PCT=90 SW_CHAIN=0,SW_SHOT_PCT=90 SW_CHAIN=1 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - SW_CHAIN=1 - > $O/knobs_64x100k.log 2>&1; cat $O/knobs_64x100k.log
timeout 200 python profiles/knob_sweep.py 256 10000000 3 -- - SW_CHAIN=1 > $O/knobs_256x10M.log 2>&1; cat $O/knobs_256x10M.log
