#!/bin/bash
# GPU call r04l: tree tally (K = 32) against the flat one on the remaining generator shapes and sizes
O=gpurun_out/r04l; mkdir -p $O
GEN_MODE=1 GEN_P0=0.02 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- - SW_TALLY_IMPL=2,SW_TALLY_K=32 > $O/knobs_cliques_256x1M.log 2>&1; cat $O/knobs_cliques_256x1M.log
GEN_MODE=3 GEN_P0=0.5 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- - SW_TALLY_IMPL=2,SW_TALLY_K=32 > $O/knobs_stale_256x1M.log 2>&1; cat $O/knobs_stale_256x1M.log
timeout 300 python profiles/knob_sweep.py 128 500000 7 -- - SW_TALLY_IMPL=2,SW_TALLY_K=32 > $O/knobs_128x500k.log 2>&1; cat $O/knobs_128x500k.log
timeout 300 python profiles/knob_sweep.py 200 800000 7 -- - SW_TALLY_IMPL=2,SW_TALLY_K=32 > $O/knobs_200x800k.log 2>&1; cat $O/knobs_200x800k.log
timeout 400 python profiles/knob_sweep.py 256 4000000 3 -- - SW_TALLY_IMPL=2,SW_TALLY_K=32 > $O/knobs_256x4M.log 2>&1; cat $O/knobs_256x4M.log
GEN_MODE=2 GEN_P0=0.1 GEN_P1=0.5 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- - SW_TALLY_IMPL=2,SW_TALLY_K=32 > $O/knobs_mild_skew_256x1M.log 2>&1; cat $O/knobs_mild_skew_256x1M.log
