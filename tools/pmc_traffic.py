#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate runs of the same
bench command, as MI355X_MICROARCH.md prescribes: TCC has 4 slots and FETCH_SIZE takes 3) into a
per-kernel HBM-traffic table.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o b -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o b -- python bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/b_counter_collection.csv \
                                gpurun_out/pmc_write/b_counter_collection.csv profiles/r01_traffic.json

Units / gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is
doubled.  The correction is calibrated here on nfs::adam_kernel (4 streamed float arrays in, 3 out: the
corrected read must equal 16 B/param) and the calibration is stored in the output.  Gather-pattern kernels
(dword loads) are uncalibrated: their read figure is an upper bound and is marked so.
"""
import collections
import csv
import json
import re
import sys

GATHER = ("rotate_render_fwd", "warp_fwd", "warp_bwd", "rotate_bwd_tiled", "p2g_", "conv3x3_c3_dgrad")


def load(path, cname):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != cname or "nfs::" not in r["Kernel_Name"]:
            continue
        short = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))
        d[short].append(float(r["Counter_Value"]))
    return d


def main():
    fetch, write, out = sys.argv[1:4]
    f = load(fetch, "FETCH_SIZE")
    w = load(write, "WRITE_SIZE")
    table = {}
    for k in sorted(set(f) | set(w)):
        # steady-state launches only: the largest-grid launches dominate; use the median of the top half
        fv = sorted(f.get(k, [0.0])); wv = sorted(w.get(k, [0.0]))
        fm = fv[len(fv) // 2:]; wm = wv[len(wv) // 2:]
        rd = 2.0 * 1024.0 * sum(fm) / len(fm)
        wr = 1024.0 * sum(wm) / len(wm)
        table[k] = {"launches_seen": len(fv), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                    # mean over ALL launches of the run (what an average-duration roofline needs)
                    "read_bytes_mean": 2.0 * 1024.0 * sum(fv) / len(fv), "write_bytes_mean": 1024.0 * sum(wv) / len(wv),
                    "read_calibrated": not any(g in k for g in GATHER)}
    cal = None
    if "nfs::adam_kernel" in table:
        t = table["nfs::adam_kernel"]
        cal = {"kernel": "nfs::adam_kernel", "read_over_write": t["read_bytes_per_launch"] / max(t["write_bytes_per_launch"], 1),
               "expected_read_over_write": 16.0 / 12.0}
    json.dump({"units": "bytes per launch (FETCH_SIZE x2 x1024, WRITE_SIZE x1024)", "calibration": cal,
               "kernels": table}, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(table.items(), key=lambda kv: -kv[1]["read_bytes_per_launch"]):
        print("%-44s read %8.1f MB  write %8.1f MB %s" % (k[:44], v["read_bytes_per_launch"] / 2 ** 20,
                                                          v["write_bytes_per_launch"] / 2 ** 20,
                                                          "" if v["read_calibrated"] else "(gather: uncalibrated)"))


if __name__ == "__main__":
    main()
