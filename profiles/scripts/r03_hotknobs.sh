#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
GEN_MODE=2 GEN_P0=0.95 GEN_P1=0.002 timeout 300 python profiles/knob_sweep.py 256 1000000 3 -- - SW_CHUNKS=8,SW_CHUNK_CFG=1,SW_PIPE=2,SW_CHUNK_MIN=8192 SW_CHUNKS=8,SW_CHUNK_CFG=1,SW_PIPE=1,SW_CHUNK_MIN=8192 SW_PIPE=2 SW_PIPE=8 SW_HALO=4096 2>&1 | tee $O/knobs_hot.log
timeout 200 python profiles/knob_sweep.py 512 1000000 3 -- - SW_SKIP=6,SW_TALLY_K=18 SW_SKIP=8,SW_TALLY_K=14 SW_SKIP=7,SW_TALLY_K=16 SW_SKIP=5,SW_TALLY_K=20 2>&1 | tee $O/knobs_512.log
