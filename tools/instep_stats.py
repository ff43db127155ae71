#!/usr/bin/env python
"""In-step per-kernel statistics from a rocprofv3 kernel trace of ``bench.py``: only the launches between the first
forward-smoothing launch of the first timed stylisation step and the end of the last step are counted (set-up work -- weight packing, the
style-target pass at B = 1, the parity case -- is excluded), so that `avg_us` of the dominant kernel is the figure
`roofline.avg_launch_us` must agree with.

    python tools/instep_stats.py <b_kernel_trace.csv> <steps> [out.json]
"""
import collections
import csv
import json
import re
import sys


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = []
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    # the forward smoothing launch opens every step (GridStylizer.forward_field; the forward advect rides in the previous step's Adam kernel since round 4)
    starts = [i for i, r in enumerate(rows) if ("smooth3d_kernel<false" in r[2])]
    if len(starts) < steps:
        raise SystemExit("found %d step starts, expected >= %d" % (len(starts), steps))
    first = starts[-steps]
    # the step after the last timed one does not exist: the last step ends with the Adam update kernel
    last = max(i for i, r in enumerate(rows) if "advect1_kernel<2" in r[2] or "adam_kernel" in r[2])
    sel = rows[first:last + 1]
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        agg[n][0] += 1
        agg[n][1] += e - s
    wall = (sel[-1][1] - sel[0][0]) / 1e6
    busy = sum(v[1] for v in agg.values()) / 1e6
    table = {n: {"launches_per_step": v[0] / steps, "avg_us": v[1] / v[0] / 1e3, "ms_per_step": v[1] / steps / 1e6}
             for n, v in agg.items()}
    gemm = [(n, v) for n, v in agg.items() if "winograd_gemm" in n]
    gl, gt = sum(v[0] for _, v in gemm), sum(v[1] for _, v in gemm)
    out = {"steps": steps, "wall_ms_per_step": wall / steps, "kernel_ms_per_step": busy / steps,
           "winograd_gemm": {"launches_per_step": gl / steps, "avg_us": gt / max(gl, 1) / 1e3,
                             "ms_per_step": gt / steps / 1e6},
           "kernels": dict(sorted(table.items(), key=lambda kv: -kv[1]["ms_per_step"]))}
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)
    print("in-step: %.3f ms/step wall, %.3f ms/step of kernel time; winograd GEMMs %.1f launches/step, avg %.1f us, "
          "%.3f ms/step" % (out["wall_ms_per_step"], out["kernel_ms_per_step"], gl / steps, gt / max(gl, 1) / 1e3,
                            gt / steps / 1e6))
    for n, v in list(out["kernels"].items())[:18]:
        print("  %-64s %6.1f /step  avg %8.1f us  %7.3f ms/step" % (n[:64], v["launches_per_step"], v["avg_us"],
                                                                   v["ms_per_step"]))


if __name__ == "__main__":
    main()
