#!/usr/bin/env python3
"""Run one multi-process GPU scenario of tests/scenarios.py and print every rank's output (development tool):
    python scripts/run_scenario.py <scenario> <ranks> ['{"json": "args"}'] [ENV=VALUE ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.gpu_harness import run_ranks  # noqa: E402

name, size = sys.argv[1], int(sys.argv[2])
args = json.loads(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].startswith("{") else {}
env = dict(kv.split("=", 1) for kv in sys.argv[3:] if "=" in kv and not kv.startswith("{"))
try:
    outs = run_ranks(name, size, args, timeout=float(os.environ.get("SCENARIO_TIMEOUT", "600")), env=env)
    print(outs[0])
except AssertionError as e:
    print(str(e)[-6000:])
    sys.exit(1)
