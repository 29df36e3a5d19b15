#!/bin/bash
# GPU call r04e: 16-bit band rows (SW_ROWS16), tree tally defaults on other shapes, find_order after slabs + bitonic median
O=gpurun_out/r04e; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_order.py tests/test_gpu_chunks.py tests/test_gpu_window.py -m gpu -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log)
tail -4 $O/pytest_subset.log
timeout 400 python profiles/knob_sweep.py 256 1000000 9 -- SW_ROWS16=0 - SW_TALLY_IMPL=2,SW_TALLY_K=32 SW_ROWS16=0,SW_TALLY_IMPL=2,SW_TALLY_K=32 SW_TALLY_IMPL=2,SW_TALLY_K=32,SW_PIPE=3 SW_TALLY_IMPL=2,SW_TALLY_K=32,SW_PIPE=5 SW_ROWS16=0 - > $O/knobs_rows16_256x1M.log 2>&1
cat $O/knobs_rows16_256x1M.log
timeout 300 python profiles/knob_sweep.py 1024 2000000 5 -- SW_ROWS16=0 - SW_TALLY_IMPL=2,SW_TALLY_K=24,SW_SKIP=6 > $O/knobs_rows16_1024x2M.log 2>&1
cat $O/knobs_rows16_1024x2M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- SW_ROWS16=0 - > $O/knobs_rows16_64x100k.log 2>&1
cat $O/knobs_rows16_64x100k.log
GEN_MODE=2 GEN_P0=0.95 GEN_P1=0.002 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- SW_ROWS16=0 - SW_TALLY_IMPL=2,SW_TALLY_K=32 > $O/knobs_hot_256x1M.log 2>&1
cat $O/knobs_hot_256x1M.log
GEN_MODE=2 GEN_P0=0.35 GEN_P1=0.02 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- SW_ROWS16=0 - SW_TALLY_IMPL=2,SW_TALLY_K=32 > $O/knobs_coin_256x1M.log 2>&1
cat $O/knobs_coin_256x1M.log
SW_PIPE=1 timeout 100 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_rows16.txt 2>&1
sed -n 1,16p $O/loop_phases_rows16.txt
SW_DEBUG_TIMING=1 timeout 300 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 2 --warmup 1 > $O/bench_order.json 2> $O/bench_order.err
grep "find_order\]" $O/bench_order.err | tail -12
python -c "
import json; d=json.load(open('$O/bench_order.json')); print('find_order_ms', d['find_order_ms'], 'ordered', d['config']['events_ordered'])"
