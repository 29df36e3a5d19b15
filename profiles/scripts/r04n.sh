#!/bin/bash
# GPU call r04n: where the round-loop iteration goes at 1024 members (146 us per iteration); tree tally at 1024 members
O=gpurun_out/r04n; mkdir -p $O
timeout 500 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_TALLY_IMPL=2,SW_TALLY_K=32 SW_TALLY_IMPL=2,SW_TALLY_K=48 SW_TALLY_IMPL=1,SW_TALLY_K=16 SW_BAND_BLOCKS=256 SW_BAND_BLOCKS=1024 > $O/knobs_1024x2M.log 2>&1; cat $O/knobs_1024x2M.log
SW_PIPE=1 SW_TALLY_IMPL=1 timeout 200 python profiles/loop_phases.py 1024 2000000 > $O/loop_phases_1024x2M.txt 2>&1; cat $O/loop_phases_1024x2M.txt
SW_TALLY_IMPL=1 timeout 200 python profiles/block_ends.py 1024 2000000 > $O/block_ends_1024x2M.txt 2>&1; cat $O/block_ends_1024x2M.txt
SW_DEBUG_CLOCKS=2 timeout 200 python profiles/resolve_time.py 1024 2000000 > $O/resolve_time_1024x2M.txt 2>&1; cat $O/resolve_time_1024x2M.txt
