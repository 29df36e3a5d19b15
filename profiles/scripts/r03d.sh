#!/bin/bash
# round 3: band-row prefetch / occupancy of k_resolve_band — parity with the knobs on, then the knob sweep
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(SW_BAND_PREFETCH=4096 SW_BAND_OCC=8 timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chunks.py -m gpu -x -q > $O/pytest_knobs_on.log 2>&1; echo "pytest rc=$?" >> $O/pytest_knobs_on.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc" $O/pytest_knobs_on.log | cut -c1-300 | tail -8
timeout 400 python profiles/knob_sweep.py 256 1000000 9 -- - SW_BAND_PREFETCH=2048 SW_BAND_PREFETCH=4096 SW_BAND_PREFETCH=8192 SW_BAND_OCC=8 SW_BAND_OCC=8,SW_BAND_PREFETCH=4096 \
   SW_BAND_OCC=8,SW_BAND_BLOCKS=256 SW_BAND_OCC=8,SW_BAND_BLOCKS=384 SW_BAND_OCC=8,SW_BAND_BLOCKS=768 SW_BAND_OCC=8,SW_BAND_BLOCKS=1024 SW_BAND_OCC=8,SW_BAND_BLOCKS=512,SW_TALLY_K=32 \
   SW_BAND_OCC=8,SW_BAND_PREFETCH=4096,SW_BAND_BLOCKS=768 SW_HALO=6144,SW_BAND_OCC=8 2>&1 | tee $O/knobs_256x1M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - SW_BAND_OCC=8 SW_BAND_PREFETCH=1024 SW_CHUNKS=1 2>&1 | tee $O/knobs_64x100k.log
