#!/bin/bash
# GPU call r04q: does a throttled finalize (fewer workgroups) disturb the round loop beside it less?  (second sweep: around 88 % / 1024)
O=gpurun_out/r04q; mkdir -p $O
timeout 800 python profiles/knob_sweep.py 256 1000000 11 -- SW_MID_PCT=88,SW_FIN_BLOCKS=1024 SW_MID_PCT=88,SW_FIN_BLOCKS=768 SW_MID_PCT=88,SW_FIN_BLOCKS=1536 SW_MID_PCT=92,SW_FIN_BLOCKS=1024 SW_MID_PCT=84,SW_FIN_BLOCKS=1024 SW_MID_PCT=88,SW_FIN_BLOCKS=512 SW_MID_PCT=94,SW_FIN_BLOCKS=1536 SW_MID_PCT=0,SW_FIN_BLOCKS=8192 SW_MID_PCT=88,SW_FIN_BLOCKS=1024 > $O/knobs2_256x1M.log 2>&1; cat $O/knobs2_256x1M.log
GEN_MODE=1 GEN_P0=0.02 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- SW_MID_PCT=0,SW_FIN_BLOCKS=8192 SW_MID_PCT=88,SW_FIN_BLOCKS=1024 > $O/knobs_cliques_256x1M.log 2>&1; cat $O/knobs_cliques_256x1M.log
timeout 300 python profiles/knob_sweep.py 256 10000000 3 -- SW_MID_PCT=0,SW_FIN_BLOCKS=8192 SW_MID_PCT=88,SW_FIN_BLOCKS=1024 > $O/knobs_256x10M.log 2>&1; cat $O/knobs_256x10M.log
timeout 300 python profiles/knob_sweep.py 64 100000 9 -- SW_MID_PCT=0,SW_FIN_BLOCKS=8192 SW_MID_PCT=88,SW_FIN_BLOCKS=1024 > $O/knobs_64x100k.log 2>&1; cat $O/knobs_64x100k.log
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- SW_MID_PCT=0,SW_FIN_BLOCKS=8192 SW_MID_PCT=88,SW_FIN_BLOCKS=1024 SW_MID_PCT=88,SW_FIN_BLOCKS=4096 > $O/knobs_1024x2M.log 2>&1; cat $O/knobs_1024x2M.log
