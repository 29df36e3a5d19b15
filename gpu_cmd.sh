SW_ELECT_IMPL=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for e in 0 1; do echo "elect $e: $(SW_ELECT_IMPL=$e timeout -k 10 300 python bench.py --cpu-sample 0 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['phase_ms']['fame'])")"; done
echo "64/100k elect 1: $(SW_ELECT_IMPL=1 timeout -k 10 300 python bench.py --members 128 --events 300000 --cpu-sample 0 | cut -c60-140)"
echo "64/100k elect 0: $(SW_ELECT_IMPL=0 timeout -k 10 300 python bench.py --members 128 --events 300000 --cpu-sample 0 | cut -c60-140)"
