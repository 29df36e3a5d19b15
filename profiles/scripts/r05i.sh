#!/bin/bash
# GPU call r05i: (side copy reverted) write-through row stores of the sweep, shot slack, the phases of an iteration along the pass
O=gpurun_out/r05i; mkdir -p $O
timeout 400 python profiles/knob_sweep.py 256 1000000 11 -- - SW_CHUNK_CFG=3 SW_SHOT_EXTRA=0 SW_SHOT_EXTRA=1 SW_CHAIN=0 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
SW_DEBUG_CLOCKS=2 timeout 100 python profiles/loop_phases.py > $O/loop_phases_along.txt 2>&1; tail -18 $O/loop_phases_along.txt
SW_DEBUG_CLOCKS=2 SW_CHUNK_CFG=3 timeout 100 python profiles/loop_phases.py > $O/loop_phases_along_wt.txt 2>&1; tail -18 $O/loop_phases_along_wt.txt
