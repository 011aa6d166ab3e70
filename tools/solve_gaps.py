"""Idle time of the device inside ONE PCG solve, from a rocprofv3 --kernel-trace CSV: the gaps between consecutive
kernels of the last complete solve phase (maximal run of PCG kernels), the largest ones listed.
Usage: python tools/solve_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
is_pcg = lambda r: any(k in r["Kernel_Name"] for k in ("k_pcg_", "k_spmv_tile", "k_tri_gemv", "copyBuffer"))
phases, cur = [], []
for r in rows:
    if is_pcg(r):
        cur.append(r)
    else:
        if len(cur) > 100:
            phases.append(cur)
        cur = []
if len(cur) > 100:
    phases.append(cur)
ph = phases[-2] if len(phases) > 1 else phases[-1]
t0, t1 = int(ph[0]["Start_Timestamp"]), int(ph[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ph)
gaps = []
end_prev = int(ph[0]["End_Timestamp"])
for i, r in enumerate(ph[1:], 1):
    s = int(r["Start_Timestamp"])
    gaps.append(((s - end_prev) / 1e3, i, r["Kernel_Name"].split("(")[0][-40:]))
    end_prev = max(end_prev, int(r["End_Timestamp"]))
print(f"solve phase: {len(ph)} kernels, {(t1 - t0) / 1e3:.1f} us wall, {busy / 1e3:.1f} us in kernels, "
      f"{sum(g for g, _, _ in gaps) / 1e3:.3f} ms of gaps")
big = sorted(gaps, reverse=True)[:12]
for g, i, n in big:
    print(f"  gap {g:7.1f} us before kernel #{i} {n}")
small = [g for g, _, _ in gaps if g < 5]
print(f"  {len(small)} gaps < 5 us, mean {sum(small) / max(1, len(small)):.2f} us")
