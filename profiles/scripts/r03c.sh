#!/bin/bash
# round 3: tests of the chunked sweep after the trip rewrite, then knob sweeps (one process each)
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(timeout 500 python -m pytest tests/test_gpu_chunks.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_chunks.log 2>&1; echo "pytest rc=$?" >> $O/pytest_chunks.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|AssertionError: can_see|pytest rc" $O/pytest_chunks.log | cut -c1-400 | tail -12
timeout 300 python profiles/knob_sweep.py 256 1000000 9 -- - SW_BAND_MAP=0 SW_CHUNKS=1 SW_CHUNKS=1,SW_BAND_MAP=0 SW_CHUNK_CFG=2 SW_CHUNKS=3 SW_CHUNKS=6,SW_CHUNK_MIN=8192 SW_HALO=6144 SW_HALO=12288 \
   SW_PIPE=1 SW_PIPE=2 SW_PIPE=3 SW_PIPE=6 SW_PIPE=8 SW_TALLY_K=32 SW_TALLY_K=30 SW_TALLY_K=24 "SW_CUTS=0.03;0.2;0.6" "SW_CUTS=0.02;0.1;0.35;0.65" "SW_CUTS=0.04;0.36;0.68" SW_BAND_BLOCKS=256 SW_BAND_BLOCKS=1024 2>&1 | tee $O/knobs_256x1M.log
