"""Idle time inside the timed steps of a rocprofv3 kernel trace: how much of the wall clock has NO kernel running (launch
gaps / dependency bubbles), and the gap that follows each kernel name on average.
    python tools/gap_stats.py <b_kernel_trace.csv> [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed steps: everything from the (steps)-th last forward-smoothing launch (first kernel of a step)
starts = [i for i, r in enumerate(rows) if ("smooth3d_kernel<false" in r["Kernel_Name"])]
first = starts[-steps]
rows = rows[first:]
t0 = int(rows[0]["Start_Timestamp"])
ev = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"]) for r in rows]
end_all = max(e for _, e, _ in ev)
busy, cur_e = 0, 0
gaps = collections.defaultdict(lambda: [0, 0])
prev_name = None
for s, e, n in ev:
    if s > cur_e:
        if prev_name is not None:
            gaps[prev_name][0] += s - cur_e
            gaps[prev_name][1] += 1
        busy += e - s
        cur_e = e
        prev_name = n
    else:
        if e > cur_e:
            busy += e - cur_e
            cur_e = e
            prev_name = n
print("wall %.3f ms/step, some kernel running %.3f ms/step, idle %.3f ms/step (%.1f %%), %d launches/step" % (
    end_all / steps / 1e6, busy / steps / 1e6, (end_all - busy) / steps / 1e6, 100.0 * (end_all - busy) / end_all, len(ev) / steps))
for n, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  idle after %-58s %6.1f us/step in %4.1f gaps/step (%.1f us each)" % (n[:58], g / steps / 1e3, c / steps, g / c / 1e3))
