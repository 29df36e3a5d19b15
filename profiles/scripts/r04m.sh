#!/bin/bash
# GPU call r04m: tree tally with the level-2 hop lists staged during level 1; automatic choice of the tally per call
O=gpurun_out/r04m; mkdir -p $O
(SW_TALLY_IMPL=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_chunks.py -m gpu -x -q > $O/pytest_tree.log 2>&1; echo "pytest rc=$?" >> $O/pytest_tree.log)
tail -4 $O/pytest_tree.log
timeout 400 python profiles/knob_sweep.py 256 1000000 9 -- - SW_TALLY_IMPL=1 SW_TALLY_IMPL=2,SW_TALLY_K=28 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
GEN_MODE=1 GEN_P0=0.02 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- - SW_TALLY_IMPL=1 > $O/knobs_cliques_256x1M.log 2>&1; cat $O/knobs_cliques_256x1M.log
GEN_MODE=2 GEN_P0=0.35 GEN_P1=0.02 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- - SW_TALLY_IMPL=2,SW_TALLY_K=32 > $O/knobs_coin_256x1M.log 2>&1; cat $O/knobs_coin_256x1M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - > $O/knobs_64x100k.log 2>&1; cat $O/knobs_64x100k.log
SW_PIPE=1 timeout 100 python profiles/loop_phases.py 256 1000000 > $O/loop_phases.txt 2>&1; sed -n 16,26p $O/loop_phases.txt
