#!/bin/bash
# GPU call r05k: do write-through row stores of the sweep shorten the kernel boundaries of the loop beside it?
O=gpurun_out/r05k; mkdir -p $O
SW_DEBUG_CLOCKS=2 SW_CHUNK_CFG=3 timeout 100 python profiles/loop_phases.py > $O/loop_phases_along_wt.txt 2>&1; tail -16 $O/loop_phases_along_wt.txt
SW_CHUNK_CFG=3 timeout 120 python profiles/subbatch_times.py > $O/subbatch_times_wt.txt 2>&1; tail -9 $O/subbatch_times_wt.txt
