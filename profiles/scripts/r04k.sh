#!/bin/bash
# GPU call r04k: the multi-slot tally (SW_TALLY_IMPL=3: four slots per wave, one workgroup per member)
O=gpurun_out/r04k; mkdir -p $O
(SW_TALLY_IMPL=3 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_chunks.py -m gpu -x -q > $O/pytest_multi.log 2>&1; echo "pytest rc=$?" >> $O/pytest_multi.log)
tail -5 $O/pytest_multi.log
timeout 400 python profiles/knob_sweep.py 256 1000000 9 -- - SW_TALLY_IMPL=3 SW_TALLY_IMPL=3,SW_TALLY_K=32 SW_TALLY_IMPL=3,SW_TALLY_K=32,SW_SKIP=0 SW_TALLY_IMPL=3,SW_TALLY_K=24 SW_TALLY_IMPL=2,SW_TALLY_K=32 - > $O/knobs_256x1M.log 2>&1
cat $O/knobs_256x1M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - SW_TALLY_IMPL=3 SW_TALLY_IMPL=3,SW_TALLY_K=32 > $O/knobs_64x100k.log 2>&1
cat $O/knobs_64x100k.log
timeout 300 python profiles/knob_sweep.py 1024 2000000 5 -- - SW_TALLY_IMPL=3 SW_TALLY_IMPL=3,SW_TALLY_K=16,SW_SKIP=8 > $O/knobs_1024x2M.log 2>&1
cat $O/knobs_1024x2M.log
GEN_MODE=2 GEN_P0=0.35 GEN_P1=0.02 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- - SW_TALLY_IMPL=3 > $O/knobs_coin_256x1M.log 2>&1
cat $O/knobs_coin_256x1M.log
GEN_MODE=2 GEN_P0=0.95 GEN_P1=0.002 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- - SW_TALLY_IMPL=3 > $O/knobs_hot_256x1M.log 2>&1
cat $O/knobs_hot_256x1M.log
SW_TALLY_IMPL=3 timeout 200 python profiles/block_ends.py 256 1000000 > $O/block_ends_multi.txt 2>&1; cat $O/block_ends_multi.txt
SW_TALLY_IMPL=3 SW_PIPE=1 timeout 100 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_multi.txt 2>&1; sed -n 16,40p $O/loop_phases_multi.txt
