#!/bin/bash
# GPU call r05n: branch-free voter loop of the elections, wide LDS scans in the tree tally; the first sweep (chunk_min / halo); parity
O=gpurun_out/r05n; mkdir -p $O
B=profiles/ab/libswirld_hip_base.so
for n in "256 1000000" "64 100000" "1024 2000000"; do
timeout 200 python profiles/fame_time.py $n 7 >> $O/fame_time.log 2>&1
done
GEN_MODE=2 GEN_P0=0.35 GEN_P1=0.02 timeout 100 python profiles/fame_time.py 256 1000000 7 >> $O/fame_time.log 2>&1; cat $O/fame_time.log
timeout 400 python profiles/knob_sweep.py 256 1000000 11 -- - SW_CHUNK_MIN=8192 SW_CHUNK_MIN=8192,SW_HALO=6144 SW_HALO=6144 SW_CHUNK_MIN=12288 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
