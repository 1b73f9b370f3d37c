"""Timeline of a rocprofv3 --kernel-trace csv: busy time per kernel and the idle gaps in front of each kernel name,
for the LAST step of tools/step_trace.py (from the last band_minmax/tile_mask burst on)."""
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# last step = after the last zonal_finalize before the final one
zi = [i for i, r in enumerate(rows) if "zonal_finalize" in r[2]]
start = zi[-2] + 1 if len(zi) >= 2 else 0
rows = rows[start:zi[-1] + 1]
t0, t1 = rows[0][0], rows[-1][1]
busy = collections.Counter(); gaps = collections.Counter(); cnt = collections.Counter(); gcnt = collections.Counter()
prev_end = rows[0][0]
for s, e, n in rows:
    k = n.split("(")[0].replace("void ", "")[:48]
    busy[k] += e - s; cnt[k] += 1
    if s > prev_end:
        gaps[k] += s - prev_end; gcnt[k] += 1
    prev_end = max(prev_end, e)
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, kernels busy {sum(busy.values()) / 1e6:.2f} ms, idle {sum(gaps.values()) / 1e6:.2f} ms, launches {len(rows)}")
for k, v in busy.most_common(30):
    print(f"{v / 1e6:8.3f} ms busy {cnt[k]:5d} calls | idle before: {gaps[k] / 1e6:7.3f} ms in {gcnt[k]:4d} gaps  {k}")
