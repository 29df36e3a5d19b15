#!/bin/bash
# round 3: where an iteration goes at HEAD (in-kernel phase stamps), the profile recipe, hot-member band knobs
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
SW_DEBUG_CLOCKS=1 SW_PIPE=1 timeout 120 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_256.txt 2>&1
SW_DEBUG_CLOCKS=1 SW_PIPE=1 timeout 120 python profiles/loop_phases.py 64 100000 > $O/loop_phases_64.txt 2>&1
head -40 $O/loop_phases_256.txt
SW_COMMIT=$(cat .commit_id 2>/dev/null || echo unknown) timeout 600 bash profiles/run_profiles.sh $1 > $O/prof.log 2>&1
cat gpurun_out/prof_$1/loop_timeline.txt | head -12
head -30 gpurun_out/prof_$1/kernel_stats.txt
GEN_MODE=2 GEN_P0=0.95 GEN_P1=0.002 timeout 300 python profiles/knob_sweep.py 256 1000000 3 -- SW_GALLOP=1 SW_GALLOP=1,SW_BAND=65536 SW_GALLOP=1,SW_BAND=262144 SW_BAND=65536 SW_GALLOP=1,SW_BAND=65536,SW_TALLY_K=16 2>&1 | tee $O/knobs_hot.log
