#!/bin/bash
# GPU call r05l: cut schedules whose second sub-batch is shorter (loop 1 waits 0.17 ms for its sweep since the loops got faster)
O=gpurun_out/r05l; mkdir -p $O
timeout 500 python profiles/knob_sweep.py 256 1000000 11 -- - "SW_CUTS=0.0625;0.2;0.43;0.71" "SW_CUTS=0.0625;0.17;0.36;0.6;0.8" "SW_CUTS=0.0625;0.15;0.3;0.5;0.75" "SW_CUTS=0.05;0.15;0.35;0.65" "SW_CUTS=0.0625;0.22;0.48;0.74" "SW_CUTS=0.04;0.12;0.3;0.53;0.76" SW_PIPE=5 SW_PIPE=6 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
