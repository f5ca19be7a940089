"""print ms_per_step and value of the last JSON line on stdin (bench.py output)"""
import json, sys
lines = [l for l in sys.stdin.read().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
print(d["ms_per_step"], d["value"])
