#!/bin/bash
# round 5 evidence at HEAD: whole GPU suite, profile recipe (kernel trace + PMC passes -> traffic.json for this kernel source), the
# default bench line, phase stamps (alone and along the pass), the other workloads of the measured table
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc" $O/pytest_gpu.log | cut -c1-300 | tail -8
SW_COMMIT=$(cat .commit_id 2>/dev/null || echo unknown) timeout 600 bash profiles/run_profiles.sh $1 > $O/prof.log 2>&1
cp gpurun_out/prof_$1/traffic.json profiles/traffic.json   # (so that the bench lines below carry PMC bytes of this code)
head -16 gpurun_out/prof_$1/kernel_stats.txt
python profiles/pass_timeline.py gpurun_out/prof_$1/kt_results.db 9 > $O/pass_timeline_256x1M.txt 2>&1; tail -1 $O/pass_timeline_256x1M.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; s=open('$O/bench_default.json').read(); d=json.loads(s[s.index('{\"metric\"'):]); print(d['value'], d['ms_per_step'], d['value_end_to_end'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['path_frac'], d['find_order_ms'], d['value_with_order'], d['value_concurrent_contexts'], d['roofline']['traffic_stale'])
for k in d['roofline']['kernels']: print(k['kernel'], k['launches'], k['avg_launch_us'], k['total_ms'], k['frac'], k['hbm_bytes_per_launch_pmc'])"
SW_DEBUG_CLOCKS=1 SW_PIPE=1 timeout 120 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_256x1M.txt 2>&1
SW_DEBUG_CLOCKS=2 timeout 120 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_along_the_pass_256x1M.txt 2>&1
SW_DEBUG_CLOCKS=2 timeout 120 python profiles/resolve_time.py 256 1000000 > $O/resolve_time_256x1M.txt 2>&1
timeout 120 python profiles/subbatch_times.py > $O/subbatch_times_256x1M.txt 2>&1
timeout 120 python profiles/fame_time.py 256 1000000 > $O/fame_time.txt 2>&1
timeout 120 python profiles/order_laps.py 256 1000000 > $O/order_laps_256x1M.txt 2>&1; tail -3 $O/order_laps_256x1M.txt
B="--cpu-sample 0 --e2e-steps 0 --contexts 1 --concurrent 0"
timeout 200 python bench.py $B --steps 5 --warmup 2 --members 1024 --events 2000000 > $O/bench_1024x2M.json 2>> $O/err.log
timeout 200 python bench.py $B --steps 10 --warmup 2 --members 64 --events 100000 > $O/bench_64x100k.json 2>> $O/err.log
timeout 300 python bench.py $B --steps 3 --warmup 1 --mode 2 --p0 0.95 --p1 0.002 > $O/bench_hot_members_256x1M.json 2>> $O/err.log
timeout 300 python bench.py $B --steps 3 --warmup 1 --mode 2 --p0 0.35 --p1 0.02 > $O/bench_coin_stress_256x1M.json 2>> $O/err.log
timeout 300 python bench.py $B --steps 3 --warmup 1 --mode 1 --p0 0.02 > $O/bench_two_cliques_256x1M.json 2>> $O/err.log
timeout 300 python bench.py $B --steps 3 --warmup 1 --mode 3 --p0 0.5 > $O/bench_stale_other_parents_256x1M.json 2>> $O/err.log
timeout 400 python bench.py $B --steps 3 --warmup 1 --events 10000000 > $O/bench_256x10M.json 2>> $O/err.log
timeout 300 python bench.py --cpu-sample 0 --e2e-steps 0 --concurrent 0 --steps 3 --warmup 1 --emulate-parts 4 > $O/bench_emulated_split.json 2>> $O/err.log
if [ "$2" = "c5" ]; then
timeout 900 python bench.py $B --steps 2 --warmup 0 --members 1024 --mode 2 --p0 0.40 --p1 0.02 --events 50000000 > $O/bench_c5_1024x50M.json 2>> $O/err.log
fi
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        s=open(f).read(); d=json.loads(s[s.index('{"metric"'):]); c=d["config"]
        print(f.split("/")[-1], "%.1f M ev/s  %.3f ms | rounds %d coin votes %d | find_order %.2f ms | strong %s" % (d["value"]/1e6, d["ms_per_step"], c["rounds"], c["coin_round_votes"], d.get("find_order_ms") or -1, json.dumps(d.get("strong"))[:160]))
    except Exception as e: print(f, "ERR", e)
PY
timeout 300 python profiles/incremental_bench.py > $O/incremental_small_batches.log 2>&1; head -2 $O/incremental_small_batches.log | cut -c1-300
find gpurun_out/prof_$1 -name '*.db' -size +4M -delete
