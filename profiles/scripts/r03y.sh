#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
timeout 600 python profiles/knob_sweep.py 1024 2000000 3 -- SW_SKIP=10,SW_TALLY_K=12 SW_SKIP=12,SW_TALLY_K=12 SW_SKIP=10,SW_TALLY_K=14 SW_SKIP=12,SW_TALLY_K=10 SW_SKIP=14,SW_TALLY_K=10 SW_SKIP=10,SW_TALLY_K=10 SW_SKIP=11,SW_TALLY_K=12 SW_SKIP=12,SW_TALLY_K=14 SW_SKIP=9,SW_TALLY_K=14 2>&1 | tee $O/knobs_1024x2M.log
GEN_MODE=2 GEN_P0=0.40 GEN_P1=0.02 timeout 600 python profiles/knob_sweep.py 1024 4000000 2 -- - SW_SKIP=10,SW_TALLY_K=12 SW_SKIP=12,SW_TALLY_K=12 SW_SKIP=8,SW_TALLY_K=16 2>&1 | tee $O/knobs_1024x4M_coin.log
GEN_MODE=1 GEN_P0=0.002 timeout 600 python profiles/knob_sweep.py 1024 4000000 2 -- - SW_SKIP=10,SW_TALLY_K=12 SW_SKIP=8,SW_TALLY_K=16 2>&1 | tee $O/knobs_1024x4M_cliques.log
timeout 300 python profiles/knob_sweep.py 512 1000000 3 -- - SW_SKIP=6,SW_TALLY_K=16 SW_SKIP=8,SW_TALLY_K=14 SW_SKIP=8,SW_TALLY_K=20 SW_SKIP=4,SW_TALLY_K=24 2>&1 | tee $O/knobs_512x1M.log
