#!/bin/bash
# GPU call r04x: a larger replay graph (fewer graph boundaries per shot)
O=gpurun_out/r04x; mkdir -p $O
timeout 600 python profiles/knob_sweep.py 256 1000000 11 -- - SW_GRAPH_BIG=48 SW_GRAPH_BIG=64 SW_GRAPH_BIG=32 SW_GRAPH_BIG=128 - > $O/knobs_graph_256x1M.log 2>&1; cat $O/knobs_graph_256x1M.log
timeout 300 python profiles/knob_sweep.py 256 10000000 3 -- - SW_GRAPH_BIG=64 SW_GRAPH_BIG=256 SW_GRAPH_BIG=512 > $O/knobs_graph_256x10M.log 2>&1; cat $O/knobs_graph_256x10M.log
SW_DEBUG_CLOCKS=1 SW_PIPE=1 SW_GRAPH_BIG=128 timeout 120 python profiles/loop_phases.py 256 1000000 2>&1 | head -3
