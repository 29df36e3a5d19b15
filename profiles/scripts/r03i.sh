#!/bin/bash
# round 3: tally workgroup size / verdict word stride, then sub-batch knobs again (the loop got faster)
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(SW_TALLY_WPB=14 SW_VSTRIDE=4 timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chunks.py -m gpu -x -q > $O/pytest_wpb14.log 2>&1; echo "pytest rc=$?" >> $O/pytest_wpb14.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc" $O/pytest_wpb14.log | cut -c1-300 | tail -5
(SW_TALLY_WPB=7 SW_VSTRIDE=16 timeout 500 python -m pytest tests/test_gpu_random.py -m gpu -x -q > $O/pytest_wpb7.log 2>&1; echo "pytest rc=$?" >> $O/pytest_wpb7.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc" $O/pytest_wpb7.log | cut -c1-300 | tail -5
timeout 400 python profiles/knob_sweep.py 256 1000000 9 -- - SW_VSTRIDE=4 SW_VSTRIDE=16 SW_TALLY_WPB=7 SW_TALLY_WPB=8 SW_TALLY_WPB=14 SW_TALLY_WPB=7,SW_VSTRIDE=4 SW_TALLY_WPB=14,SW_VSTRIDE=4 \
    SW_PIPE=3 SW_PIPE=5 SW_PIPE=6 SW_PIPE=8 SW_CHUNKS=3 SW_CHUNKS=5,SW_CHUNK_MIN=8192 SW_CHUNKS=8,SW_CHUNK_MIN=8192 SW_HALO=6144 SW_BAND_BLOCKS=384 SW_BAND_BLOCKS=640 2>&1 | tee $O/knobs_256x1M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - SW_TALLY_WPB=7 SW_TALLY_WPB=14 SW_VSTRIDE=16 SW_BAND_BLOCKS=128 SW_BAND_BLOCKS=256 2>&1 | tee $O/knobs_64x100k.log
