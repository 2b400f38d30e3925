"""Per-launch durations of the render kernel from a rocprofv3 kernel trace (CSV), in launch order,
grouped in runs of `group` launches: min of each group in microseconds.
  python tools/trace_durations.py <dir with *_kernel_trace.csv> [group]"""
import csv, glob, sys
group = int(sys.argv[2]) if len(sys.argv) > 2 else 13
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
rows.sort()
for i in range(0, len(rows), group):
    g = rows[i:i + group]
    d = [(e - s) / 1e3 for s, e, *_ in g]
    gaps = [(g[k + 1][0] - g[k][1]) / 1e3 for k in range(len(g) - 1)]
    print(f"launches {i:4d}..{i + len(g) - 1:4d}: kernel min {min(d):8.1f} us  median {sorted(d)[len(d) // 2]:8.1f} us   gap to next launch min {min(gaps) if gaps else 0:7.1f} us   grid {g[0][3]} wg {g[0][4]} {g[0][2]}")
