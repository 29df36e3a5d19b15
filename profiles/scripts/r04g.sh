#!/bin/bash
# GPU call r04g: host reorderings (deferred bounds sync, aux launches behind the next loop's first shot), SW_SWEEP_NAP,
# then the WHOLE GPU suite (new 1024-member coin-round oracle case included)
O=gpurun_out/r04g; mkdir -p $O
timeout 400 python profiles/knob_sweep.py 256 1000000 9 -- - SW_SWEEP_NAP=1 SW_TALLY_IMPL=2,SW_TALLY_K=32 SW_TALLY_IMPL=2,SW_TALLY_K=32,SW_SWEEP_NAP=1 - SW_SWEEP_NAP=1 > $O/knobs_256x1M.log 2>&1
cat $O/knobs_256x1M.log
GEN_MODE=2 GEN_P0=0.95 GEN_P1=0.002 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- - SW_SWEEP_NAP=1 > $O/knobs_hot_256x1M.log 2>&1
cat $O/knobs_hot_256x1M.log
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log)
tail -6 $O/pytest_gpu.log
