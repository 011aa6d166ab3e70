"""One-line digest of a bench.py JSON line read from stdin (label as first argument)."""
import json
import sys

d = json.loads(sys.stdin.read())
k = d["kernels_ms"]
print(sys.argv[1] if len(sys.argv) > 1 else "", f"{d['value'] / 1e6:.1f} M beams/s", f"{d['ms_per_step']:.2f} ms/step",
      d["config"]["pcg_iterations"], "its | assembly", round(k["assembly_ms_last"], 2), "solve", round(k["solve_ms_last"], 2),
      "| K*p", round(k["spmv"] * 1e3, 1), "iteration", round(k["pcg_iteration"] * 1e3, 1), "us")
