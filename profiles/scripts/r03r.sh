#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
GEN_MODE=2 GEN_P0=0.95 GEN_P1=0.002 timeout 300 python profiles/knob_sweep.py 256 1000000 3 -- - SW_CHUNK_CFG=2 SW_CHUNKS=8,SW_CHUNK_MIN=8192,SW_CHUNK_CFG=2 SW_CHUNKS=6,SW_CHUNK_MIN=8192,SW_CHUNK_CFG=2 SW_CHUNKS=8,SW_CHUNK_MIN=8192,SW_CHUNK_CFG=2,SW_HALO=4096 SW_CHUNKS=8,SW_CHUNK_MIN=8192,SW_CHUNK_CFG=1 2>&1 | tee $O/knobs_hot.log
timeout 200 python profiles/knob_sweep.py 256 1000000 5 -- - SW_CHUNK_CFG=2 SW_CHUNKS=8,SW_CHUNK_MIN=8192,SW_CHUNK_CFG=2 SW_CHUNKS=6,SW_CHUNK_MIN=8192,SW_CHUNK_CFG=2 2>&1 | tee $O/knobs_uniform.log
