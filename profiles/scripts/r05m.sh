#!/bin/bash
# GPU call r05m: graduated sub-batch schedule (default now) against the even one, across shapes; then the whole GPU suite
O=gpurun_out/r05m; mkdir -p $O
OLD="SW_CUTS=0.0625;0.297;0.531;0.766"
timeout 300 python profiles/knob_sweep.py 256 1000000 11 -- - "$OLD" SW_PIPE=4 SW_PIPE=6 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - "$OLD" - > $O/knobs_64x100k.log 2>&1; cat $O/knobs_64x100k.log
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - "$OLD" > $O/knobs_1024x2M.log 2>&1; cat $O/knobs_1024x2M.log
timeout 200 python profiles/knob_sweep.py 256 10000000 3 -- - "$OLD" > $O/knobs_256x10M.log 2>&1; cat $O/knobs_256x10M.log
GEN_MODE=1 GEN_P0=0.02 timeout 200 python profiles/knob_sweep.py 256 1000000 5 -- - "$OLD" > $O/knobs_cliques_256x1M.log 2>&1; cat $O/knobs_cliques_256x1M.log
GEN_MODE=2 GEN_P0=0.35 GEN_P1=0.02 timeout 200 python profiles/knob_sweep.py 256 1000000 5 -- - "$OLD" > $O/knobs_coin_256x1M.log 2>&1; cat $O/knobs_coin_256x1M.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
