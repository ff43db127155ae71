"""rocprofv3 --kernel-trace CSV -> average duration per (kernel, grid size): separates the shapes one kernel runs at."""
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0][:60]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    acc[(name, r.get("Grid_Size_X", r.get("Grid_Size", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), v in sorted(acc.items()):
    v = sorted(v)
    print("%-62s grid %8s  n=%4d  median %8.1f us  min %8.1f" % (name, grid, len(v), v[len(v) // 2], v[0]))
