#!/bin/bash
# GPU call r05a: the two compile-checked candidates of round 4 (band fast path for full groups; occupancy of the
# 1024-member tally) and streaming stores for the sweep's rows — knob sweeps, then parity of the winners
O=gpurun_out/r05a; mkdir -p $O
timeout 300 python profiles/knob_sweep.py 256 1000000 11 -- - SW_BAND_FAST=1 SW_CHUNK_CFG=3 SW_BAND_FAST=1,SW_CHUNK_CFG=3 - SW_BAND_FAST=1 > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
SW_BAND_FAST=0 timeout 100 python profiles/resolve_time.py > $O/resolve_time_base.txt 2>&1; cat $O/resolve_time_base.txt
SW_BAND_FAST=1 timeout 100 python profiles/resolve_time.py > $O/resolve_time_fast.txt 2>&1; cat $O/resolve_time_fast.txt
timeout 500 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_TALLY_MINW=6 SW_TALLY_MINW=5 - > $O/knobs_1024x2M.log 2>&1; cat $O/knobs_1024x2M.log
GEN_MODE=1 GEN_P0=0.02 timeout 200 python profiles/knob_sweep.py 256 1000000 5 -- - SW_BAND_FAST=1 SW_CHUNK_CFG=3 - > $O/knobs_cliques_256x1M.log 2>&1; cat $O/knobs_cliques_256x1M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - SW_BAND_FAST=1 - > $O/knobs_64x100k.log 2>&1; cat $O/knobs_64x100k.log
SW_BAND_FAST=1 SW_CHUNK_CFG=3 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_chunks.py -m gpu -x -q > $O/pytest_fast.log 2>&1; tail -3 $O/pytest_fast.log
