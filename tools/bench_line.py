"""One-line digest of a bench.py JSON line read from stdin (label as first argument)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get("kernels_ms", {})
c = d.get("config", {})
parts = [sys.argv[1] if len(sys.argv) > 1 else "", f"{d['value'] / 1e6:.1f} M beams/s", f"{d['ms_per_step']:.2f} ms/step"]
if "pcg_iterations" in c:
    parts += [str(c["pcg_iterations"]), "its"]
if "assembly_ms_last" in k:      # (the design-loop configuration reports per design iteration and has no kernel timings)
    parts += ["| assembly", str(round(k["assembly_ms_last"], 2)), "solve", str(round(k["solve_ms_last"], 2)),
              "| K*p", str(round(k["spmv"] * 1e3, 1)), "iteration", str(round(k["pcg_iteration"] * 1e3, 1)), "us"]
print(" ".join(parts))
