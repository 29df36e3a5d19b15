#!/bin/bash
# GPU call r04d: tree tally after the forced-slot fix (parity), band-row prefetch wave (SW_BAND_PRE), find_order after the
# regrouped first-descendant kernel and the device-side segment gather
O=gpurun_out/r04d; mkdir -p $O
(SW_TALLY_IMPL=2 SW_BAND_PRE=6144 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_order.py -m gpu -x -q > $O/pytest_tree_pf.log 2>&1; echo "pytest rc=$?" >> $O/pytest_tree_pf.log)
tail -4 $O/pytest_tree_pf.log
timeout 400 python profiles/knob_sweep.py 256 1000000 9 -- - SW_BAND_PRE=4096 SW_BAND_PRE=6144 SW_BAND_PRE=8192 SW_BAND_PRE=6144,SW_TALLY_IMPL=2,SW_TALLY_K=32 SW_BAND_PRE=6144,SW_TALLY_IMPL=2,SW_TALLY_K=32,SW_SKIP=0 SW_BAND_PRE=12288 SW_BAND_PRE=6144,SW_BAND_BLOCKS=384 SW_BAND_PRE=6144,SW_BAND_BLOCKS=768 - > $O/knobs_pf_256x1M.log 2>&1
cat $O/knobs_pf_256x1M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - SW_BAND_PRE=1024 SW_BAND_PRE=2048 SW_BAND_PRE=4096 > $O/knobs_pf_64x100k.log 2>&1
cat $O/knobs_pf_64x100k.log
SW_BAND_PRE=6144 SW_PIPE=1 timeout 100 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_pf.txt 2>&1
head -16 $O/loop_phases_pf.txt
SW_DEBUG_TIMING=1 timeout 300 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 2 --warmup 1 > $O/bench_order.json 2> $O/bench_order.err
grep "find_order\]" $O/bench_order.err | tail -12
python -c "
import json; d=json.load(open('$O/bench_order.json')); print('find_order_ms', d['find_order_ms'], 'ordered', d['config']['events_ordered'])"
