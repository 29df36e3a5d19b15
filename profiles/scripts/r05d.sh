#!/bin/bash
# GPU call r05d: wave reductions and the bit-sliced adders' cross-lane steps by DPP + v_permlane16/32_swap instead of ds_bpermute;
# v_readlane for the tree tally's slot look-ups; plain launches against graph replays now that an iteration is ~16 us; flat against
# two-level tally again.  (calibration: the build of commit 126ad6c, 6.34-6.36 ms in r05b / r05c)
O=gpurun_out/r05d; mkdir -p $O
B=profiles/ab/libswirld_hip_base.so
SWEEP_LIB=$B timeout 200 python profiles/knob_sweep.py 256 1000000 11 -- - > $O/ab_256x1M.log 2>&1
timeout 300 python profiles/knob_sweep.py 256 1000000 11 -- - SW_GRAPH=0 SW_TALLY_IMPL=1,SW_TALLY_K=28 SW_TALLY_IMPL=1,SW_TALLY_K=28,SW_GRAPH=0 SW_TALLY_IMPL=1,SW_TALLY_K=32 - SW_GRAPH=0 >> $O/ab_256x1M.log 2>&1; cat $O/ab_256x1M.log
timeout 100 python profiles/resolve_time.py > $O/resolve_time.txt 2>&1; cat $O/resolve_time.txt
timeout 100 python profiles/loop_phases.py > $O/loop_phases.txt 2>&1; sed -n 1,24p $O/loop_phases.txt
SWEEP_LIB=$B timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - > $O/ab_64x100k.log 2>&1
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - SW_GRAPH=0 - SW_GRAPH=0 >> $O/ab_64x100k.log 2>&1; cat $O/ab_64x100k.log
SWEEP_LIB=$B timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - > $O/ab_1024x2M.log 2>&1
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_GRAPH=0 >> $O/ab_1024x2M.log 2>&1; cat $O/ab_1024x2M.log
GEN_MODE=1 GEN_P0=0.02 timeout 200 python profiles/knob_sweep.py 256 1000000 5 -- - SW_GRAPH=0 - > $O/knobs_cliques_256x1M.log 2>&1; cat $O/knobs_cliques_256x1M.log
timeout 200 python profiles/knob_sweep.py 256 10000000 3 -- - SW_GRAPH=0 > $O/knobs_256x10M.log 2>&1; cat $O/knobs_256x10M.log
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --e2e-steps 0 --contexts 1 > $O/bench_graph.json 2> $O/bench_graph.err; python -c "
import json;d=json.loads(open('$O/bench_graph.json').read().strip().splitlines()[-1]);print('bench graph', d['value'], d['ms_per_step'])"
SW_GRAPH=0 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --e2e-steps 0 --contexts 1 > $O/bench_plain.json 2> $O/bench_plain.err; python -c "
import json;d=json.loads(open('$O/bench_plain.json').read().strip().splitlines()[-1]);print('bench plain', d['value'], d['ms_per_step'])"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_partition.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
