"""Timeline of ONE pl_assemble + pl_assemble_bsr from a rocprofv3 --kernel-trace CSV: every kernel between the last
PCG kernel of one step and the first PCG kernel of the next, with start offsets, durations and the gaps on the device.
Usage: python tools/assembly_timeline.py <kernel_trace.csv> [step index from the end = 2]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pl::", "")
is_pcg = lambda r: any(k in r["Kernel_Name"] for k in ("k_pcg_", "k_spmv_tile", "k_tri_gemv", "k_full_gemv", "k_small_z"))
# assembly phases = maximal runs of non-PCG kernels that contain k_build_records
phases, cur = [], []
for r in rows:
    if is_pcg(r):
        if cur and any("record" in name(x) for x in cur):
            phases.append(cur)
        cur = []
    else:
        cur.append(r)
ph = phases[-back]
t0 = int(ph[0]["Start_Timestamp"])
end_prev = t0
busy = 0
print(f"{'start us':>9} {'dur us':>8} {'gap us':>7}  kernel")
for r in ph:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - end_prev) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {name(r)[:70]}")
    end_prev = max(end_prev, e)
print(f"phase length {(end_prev - t0) / 1e3:.1f} us, {len(ph)} kernels")
