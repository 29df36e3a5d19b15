"""CPU: bench.py's guard around the one-hashgraph split (world > 1): work that blocks for ever inside a collective must cost
the run `value_strong`, never the JSON line — the line is printed from a timer thread and the process ends with status 0."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, time, threading
sys.path.insert(0, %r)
import bench
lock, done = threading.Lock(), [False]
def emit(x):
    with lock:
        if done[0]:
            return
        done[0] = True
    print("LINE %%s" %% x, flush=True)
mode = sys.argv[1]
def work():
    if mode == "hang":
        threading.Event().wait()          # a collective that never completes
    if mode == "raise":
        raise RuntimeError("boom")
    time.sleep(0.2)
    return "result"
def run():
    try:
        return work()
    except Exception as exc:
        return "error %%r" %% (exc,)
r = bench.guarded(run, 1.0 if mode != "nolimit" else 0, lambda: emit("timeout"), linger_s=0.2)
emit(r)
print("after", flush=True)
"""


def run(mode):
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", SCRIPT % ROOT, mode], capture_output=True, text=True, timeout=60)
    return p.returncode, p.stdout.strip().splitlines(), time.time() - t0


def test_guard_returns_the_result_and_cancels_the_timer():
    rc, out, dt = run("ok")
    assert rc == 0 and out == ["LINE result", "after"]
    rc, out, dt = run("nolimit")
    assert rc == 0 and out == ["LINE result", "after"]


def test_guard_passes_exceptions_on_as_results():
    rc, out, dt = run("raise")
    assert rc == 0 and out[0].startswith("LINE error") and "boom" in out[0] and out[1] == "after"


def test_guard_prints_the_line_and_ends_the_process_when_the_work_hangs():
    rc, out, dt = run("hang")
    assert rc == 0 and out == ["LINE timeout"] and dt < 30   # (exactly one line: nothing after the hang runs)
