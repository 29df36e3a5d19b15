#!/bin/bash
# GPU call r05v: last knob checks at the final kernels (band workgroups, elections tile, graph size), parity of the touched tests
O=gpurun_out/r05v; mkdir -p $O
timeout 400 python profiles/knob_sweep.py 256 1000000 11 -- - SW_BAND_BLOCKS=384 SW_BAND_BLOCKS=768 SW_BAND_BLOCKS=1024 SW_ELECT_CG=64 SW_ELECT_CG=256 SW_GRAPH_BIG=128 SW_TALLY_PF=0 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
SW_ELECT_CG=64 timeout 100 python profiles/fame_time.py 256 1000000 7 > $O/fame_cg64.log 2>&1; cat $O/fame_cg64.log
timeout 100 python profiles/fame_time.py 256 1000000 7 > $O/fame_cg128.log 2>&1; cat $O/fame_cg128.log
timeout 600 python -m pytest tests/test_gpu_errors.py tests/test_gpu_random.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
