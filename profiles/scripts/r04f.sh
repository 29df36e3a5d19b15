#!/bin/bash
# GPU call r04f: band phase with one vector load per row (band mask layout), with and without the 16-bit side table;
# the windowed-table test that aborted in r04e, on its own with stderr; find_order second-call timing
O=gpurun_out/r04f; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_window.py -m gpu -x -q -s > $O/pytest_window.log 2>&1; echo "pytest rc=$?" >> $O/pytest_window.log)
tail -6 $O/pytest_window.log
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_order.py tests/test_gpu_chunks.py -m gpu -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log)
tail -4 $O/pytest_subset.log
timeout 400 python profiles/knob_sweep.py 256 1000000 9 -- SW_ROWS16=0 - SW_ROWS16=0,SW_TALLY_IMPL=2,SW_TALLY_K=32 SW_TALLY_IMPL=2,SW_TALLY_K=32 SW_ROWS16=0,SW_BAND_BLOCKS=384 SW_ROWS16=0,SW_BAND_BLOCKS=256 SW_BAND_BLOCKS=256 SW_ROWS16=0 - > $O/knobs_vec_256x1M.log 2>&1
cat $O/knobs_vec_256x1M.log
timeout 300 python profiles/knob_sweep.py 1024 2000000 5 -- SW_ROWS16=0 - > $O/knobs_vec_1024x2M.log 2>&1
cat $O/knobs_vec_1024x2M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- SW_ROWS16=0 - > $O/knobs_vec_64x100k.log 2>&1
cat $O/knobs_vec_64x100k.log
GEN_MODE=2 GEN_P0=0.35 GEN_P1=0.02 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- SW_ROWS16=0 - > $O/knobs_vec_coin_256x1M.log 2>&1
cat $O/knobs_vec_coin_256x1M.log
SW_ROWS16=0 SW_PIPE=1 timeout 100 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_vec32.txt 2>&1
sed -n 1,16p $O/loop_phases_vec32.txt
SW_PIPE=1 timeout 100 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_vec16.txt 2>&1
sed -n 12,15p $O/loop_phases_vec16.txt
timeout 300 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 2 --warmup 1 > $O/bench_order.json 2> $O/bench_order.err
python -c "
import json; d=json.load(open('$O/bench_order.json')); print('find_order_ms', d['find_order_ms'], 'first', d['find_order_first_call_ms'], 'value', d['value'])"
