#!/bin/bash
# GPU call r05j: the phases of an iteration ALONG the pass (what the sweeps beside the first loops cost, and where)
O=gpurun_out/r05j; mkdir -p $O
SW_DEBUG_CLOCKS=2 timeout 100 python profiles/loop_phases.py > $O/loop_phases_along.txt 2>&1; tail -20 $O/loop_phases_along.txt
SW_DEBUG_CLOCKS=2 SW_PIPE=1 timeout 100 python profiles/loop_phases.py > $O/loop_phases_along_pipe1.txt 2>&1; tail -16 $O/loop_phases_along_pipe1.txt
timeout 200 python profiles/knob_sweep.py 256 1000000 11 -- - SW_CHAIN=1 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
