#!/bin/bash
# GPU call r05c: compile-time wave count + 32-bit offsets in the resolve step (reductions no longer remainder loops), bracket-row
# touch of the two-level tally (SW_TALLY_PF=3), graph packet capture / graph sizes (is the host the limit?) — against the build of
# commit 126ad6c on the same box
O=gpurun_out/r05c; mkdir -p $O
B=profiles/ab/libswirld_hip_base.so
for i in 1 2; do
SWEEP_LIB=$B timeout 200 python profiles/knob_sweep.py 256 1000000 11 -- - >> $O/ab_256x1M.log 2>&1
timeout 200 python profiles/knob_sweep.py 256 1000000 11 -- - SW_TALLY_PF=3 SW_TALLY_PF=0 SW_TALLY_PF=2 - >> $O/ab_256x1M.log 2>&1
done; cat $O/ab_256x1M.log
timeout 100 python profiles/resolve_time.py > $O/resolve_time.txt 2>&1; cat $O/resolve_time.txt
SW_TALLY_PF=3 timeout 100 python profiles/loop_phases.py > $O/loop_phases_pf3.txt 2>&1; sed -n 1,30p $O/loop_phases_pf3.txt
timeout 100 python profiles/loop_phases.py > $O/loop_phases.txt 2>&1; sed -n 16,24p $O/loop_phases.txt
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 timeout 200 python profiles/knob_sweep.py 256 1000000 9 -- - SW_GRAPH_BIG=128 SW_GRAPH_BIG=256 > $O/graph_capture1.log 2>&1; cat $O/graph_capture1.log
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 200 python profiles/knob_sweep.py 256 1000000 9 -- - SW_GRAPH_BIG=128 SW_GRAPH=0 > $O/graph_capture0.log 2>&1; cat $O/graph_capture0.log
SWEEP_LIB=$B timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - > $O/ab_64x100k.log 2>&1
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - - >> $O/ab_64x100k.log 2>&1; cat $O/ab_64x100k.log
SWEEP_LIB=$B timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - > $O/ab_1024x2M.log 2>&1
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - >> $O/ab_1024x2M.log 2>&1; cat $O/ab_1024x2M.log
SW_TALLY_PF=3 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
