"""Timeline of ONE graph replay of Depth-Anything-v3 small at 518^2 (BASELINE config 2) from a rocprofv3 --kernel-trace CSV:
per launch start / duration / gap to the previous end on the same queue, and the union of busy intervals against the wall time.
usage: python tools/probes/cfg2_timeline.py <kernel_trace.csv> [launches_per_replay]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last replay: walk back from the end until the first kernel of an infer (patchify) is seen
idx = [i for i, r in enumerate(rows) if "patchify" in r["Kernel_Name"]]
lo = idx[-1]
hi = len(rows)
rep = rows[lo:hi]
t0 = int(rep[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in rep)
print(f"{len(rep)} launches, wall {(end - t0) / 1e3:.1f} us")
busy, cur_s, cur_e = 0, None, None
for r in rep:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"union of kernel intervals {busy / 1e3:.1f} us ({100.0 * busy / (end - t0):.1f} % of the wall), sum of durations {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rep) / 1e3:.1f} us")
prev_end = t0
fam = {}
for r in rep:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void md::", "").replace("md::", "")[:60]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:7.2f}  wg {r.get('Workgroup_Size', '?'):>5} grid {r.get('Grid_Size', '?'):>8}  {name}")
    prev_end = max(prev_end, e)
    f = fam.setdefault(name, [0, 0.0])
    f[0] += 1
    f[1] += (e - s) / 1e3
print("--- by kernel ---")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]:8.1f} us  {v[0]:3d} x  {k}")
