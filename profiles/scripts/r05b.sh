#!/bin/bash
# GPU call r05b: ISA-level clean-ups of the loop kernels (creator load of the band groups no longer waited for in front of the rows;
# unconditional clamped loads at the head of the resolve step; the tally's member words requested before its band-table touch)
# against the build of r05a (same box, alternating processes), then parity
O=gpurun_out/r05b; mkdir -p $O
B=profiles/ab/libswirld_hip_base.so
for i in 1 2; do
SWEEP_LIB=$B timeout 200 python profiles/knob_sweep.py 256 1000000 11 -- SW_BAND_FAST=1 SW_BAND_FAST=1 >> $O/ab_256x1M.log 2>&1
timeout 200 python profiles/knob_sweep.py 256 1000000 11 -- - SW_BAND_FAST=0 - >> $O/ab_256x1M.log 2>&1
done; cat $O/ab_256x1M.log
timeout 100 python profiles/resolve_time.py > $O/resolve_time.txt 2>&1; cat $O/resolve_time.txt
timeout 100 python profiles/loop_phases.py > $O/loop_phases.txt 2>&1; head -45 $O/loop_phases.txt
SWEEP_LIB=$B timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - > $O/ab_1024x2M.log 2>&1
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - - >> $O/ab_1024x2M.log 2>&1; cat $O/ab_1024x2M.log
SWEEP_LIB=$B SW_BAND_FAST=1 timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - > $O/ab_64x100k.log 2>&1
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - - >> $O/ab_64x100k.log 2>&1; cat $O/ab_64x100k.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_chunks.py tests/test_gpu_errors.py tests/test_gpu_node.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
