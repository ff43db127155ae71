"""Kernel timeline of ONE step from a rocprofv3 kernel trace: start offset, duration and the idle gap before every launch
(the step = from the last-but-one forward-smoothing launch to the next one; any other kernel-name fragment marks the
first launch of a step of another loop, e.g. p2g_fwd for the particle stylers).
    python tools/timeline.py <kernel_trace.csv> [step_from_end=2] [first-kernel-of-a-step=smooth3d_kernel<false]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[3] if len(sys.argv) > 3 else "smooth3d_kernel<false"
starts = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
a, b = starts[-k - 1], starts[-k]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot_busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("nfs::", ""))
    print("%9.1f us  +%6.1f gap  %7.1f us  %s  grid %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name[:70],
                                                          r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    prev_end = max(prev_end, e)
    tot_busy += e - s
print("step: %.1f us wall, %.1f us of kernels, %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, tot_busy / 1e3, b - a))
