#!/bin/bash
# GPU call r04zb: three ways for the band pass to write round[] / S[] (plain, write-through, creators fetched behind the rows)
O=gpurun_out/r04zb; mkdir -p $O
timeout 500 python profiles/knob_sweep.py 256 1000000 11 -- - SW_FIN_VAR=1 SW_FIN_VAR=2 - SW_FIN_VAR=1 SW_FIN_VAR=2 > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
for v in 0 1 2; do SW_FIN_VAR=$v SW_DEBUG_CLOCKS=2 timeout 120 python profiles/resolve_time.py 256 1000000 2>&1 | tail -3; done
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_FIN_VAR=1 SW_FIN_VAR=2 > $O/knobs_1024x2M.log 2>&1; cat $O/knobs_1024x2M.log
