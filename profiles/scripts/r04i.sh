#!/bin/bash
# GPU call r04i: resolve step with the commit test and the next round's counts on one barrier, inheritance pass skipped
# unless a tally flagged a FAR candidate
O=gpurun_out/r04i; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_chunks.py tests/test_gpu_ingest.py tests/test_gpu_node.py -m gpu -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log)
tail -4 $O/pytest_subset.log
timeout 400 python profiles/knob_sweep.py 256 1000000 9 -- - SW_TALLY_IMPL=2,SW_TALLY_K=32 - > $O/knobs_256x1M.log 2>&1
cat $O/knobs_256x1M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - > $O/knobs_64x100k.log 2>&1
cat $O/knobs_64x100k.log
GEN_MODE=2 GEN_P0=0.35 GEN_P1=0.02 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- - > $O/knobs_coin_256x1M.log 2>&1
cat $O/knobs_coin_256x1M.log
SW_DEBUG_CLOCKS=2 timeout 100 python profiles/resolve_time.py 256 1000000 > $O/resolve_time.txt 2>&1
cat $O/resolve_time.txt
