#!/bin/bash
# GPU call r04w: untraced per-sub-batch loop times (is the first loop really slower beside the sweeps?)
O=gpurun_out/r04w; mkdir -p $O
timeout 200 python profiles/subbatch_times.py 256 1000000 > $O/subbatch_times_256x1M.txt 2>&1; tail -32 $O/subbatch_times_256x1M.txt
SW_PIPE=1 timeout 200 python profiles/subbatch_times.py 256 1000000 > $O/subbatch_times_256x1M_pipe1.txt 2>&1; tail -8 $O/subbatch_times_256x1M_pipe1.txt
