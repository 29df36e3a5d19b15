timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do timeout -k 10 300 python bench.py --cpu-sample 0 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['launches'])"; done
timeout -k 10 300 python bench.py --members 64 --events 100000 --cpu-sample 0 | cut -c60-140
SW_PIPE=1 timeout -k 10 300 python profiles/loop_phases.py 256 1000000 | head -8
