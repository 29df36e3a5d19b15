#!/bin/bash
# GPU call r04y: which iterations of the round loop are the slow ones (p90 26 us against a median of 18)?
O=gpurun_out/r04y; mkdir -p $O
SW_DEBUG_CLOCKS=1 SW_PIPE=1 timeout 120 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_256x1M.txt 2>&1; tail -9 $O/loop_phases_256x1M.txt
SW_DEBUG_CLOCKS=1 SW_PIPE=1 SW_TALLY_IMPL=1 timeout 120 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_256x1M_flat_tally.txt 2>&1; tail -9 $O/loop_phases_256x1M_flat_tally.txt
