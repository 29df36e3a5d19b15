#!/bin/bash
# GPU call r04zc: 1024 members with the band finalize: window, offset and workgroup count revisited
O=gpurun_out/r04zc; mkdir -p $O
timeout 700 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_TALLY_K=8,SW_SKIP=10 SW_TALLY_K=16,SW_SKIP=10 SW_TALLY_K=12,SW_SKIP=8 SW_TALLY_K=12,SW_SKIP=12 SW_TALLY_K=8,SW_SKIP=12 SW_BAND_BLOCKS=192 SW_BAND_BLOCKS=384 SW_GALLOP=0 SW_PIPE=3 SW_PIPE=6 - > $O/knobs_1024x2M.log 2>&1; cat $O/knobs_1024x2M.log
