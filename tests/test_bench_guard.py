"""bench.py's LineGuard: stdout carries exactly ONE JSON line even when the bench process dies natively after the timed pass (an abort
inside RCCL's watchdog or a profiler child is not a Python exception).  CPU-only: the guard is plain fork + pipe."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import bench
fd = os.dup(1); os.dup2(2, 1)
bench._JSON_FD = fd
bench._GUARD = bench.LineGuard(fd)
print("noise on fd 1 goes to stderr")
bench._GUARD.send(json.dumps({{"value": 1, "note": "provisional"}}))
bench._GUARD.send(json.dumps({{"value": 2, "note": "provisional, after a leg", "pad": "x" * 3000}}))
mode = sys.argv[1]
if mode == "abort":
    os.abort()
if mode == "kill":
    os.kill(os.getpid(), 9)
if mode == "exit":
    os._exit(3)
bench.emit_line(json.dumps({{"value": 3, "note": "final"}}))
bench.emit_line(json.dumps({{"value": 4, "note": "a second emit_line without a guard writes directly: never happens in bench.py"}})) if mode == "twice" else None
"""


def _run(mode):
    p = subprocess.run([sys.executable, "-c", PROG.format(root=ROOT), mode], capture_output=True, text=True, timeout=120)
    return p.returncode, [l for l in p.stdout.splitlines() if l.strip()]


def test_final_line_wins_when_the_bench_finishes():
    rc, lines = _run("final")
    assert rc == 0 and len(lines) == 1 and json.loads(lines[0]) == {"value": 3, "note": "final"}


def test_last_provisional_line_is_printed_when_the_process_dies():
    for mode in ("abort", "kill", "exit"):
        rc, lines = _run(mode)
        assert rc != 0 and len(lines) == 1, (mode, rc, lines)
        d = json.loads(lines[0])
        assert d["value"] == 2 and len(d["pad"]) == 3000, mode
