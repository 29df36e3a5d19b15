#!/bin/bash
# GPU call r04c: the two-level (tree) tally against one wave per slot; parity of the tree under the suite's
# heaviest cases; find_order stage clocks of the bulk path
O=gpurun_out/r04c; mkdir -p $O
(SW_TALLY_IMPL=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_chunks.py -m gpu -x -q > $O/pytest_tree.log 2>&1; echo "pytest rc=$?" >> $O/pytest_tree.log)
tail -4 $O/pytest_tree.log
timeout 300 python profiles/knob_sweep.py 256 1000000 9 -- - SW_TALLY_IMPL=2 SW_TALLY_IMPL=2,SW_TALLY_K=32 SW_TALLY_IMPL=2,SW_TALLY_K=24 SW_TALLY_IMPL=2,SW_TALLY_K=40,SW_SKIP=0 SW_TALLY_IMPL=2,SW_TALLY_K=32,SW_SKIP=0 SW_TALLY_IMPL=2,SW_TALLY_K=48 - > $O/knobs_tree_256x1M.log 2>&1
cat $O/knobs_tree_256x1M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - SW_TALLY_IMPL=2 SW_TALLY_IMPL=2,SW_TALLY_K=32 > $O/knobs_tree_64x100k.log 2>&1
cat $O/knobs_tree_64x100k.log
timeout 300 python profiles/knob_sweep.py 1024 2000000 5 -- - SW_TALLY_IMPL=2 SW_TALLY_IMPL=2,SW_TALLY_K=16,SW_SKIP=8 SW_TALLY_IMPL=2,SW_TALLY_K=24,SW_SKIP=6 > $O/knobs_tree_1024x2M.log 2>&1
cat $O/knobs_tree_1024x2M.log
SW_TALLY_IMPL=2 SW_PIPE=1 timeout 100 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_tree.txt 2>&1
SW_PIPE=1 timeout 100 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_flat.txt 2>&1
head -40 $O/loop_phases_tree.txt
SW_DEBUG_TIMING=1 timeout 300 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 2 --warmup 1 > $O/bench_order.json 2> $O/bench_order.err
grep "find_order\]" $O/bench_order.err | tail -12
