"""dev: kernel timeline of the last steps in a rocprofv3 kernel trace (any config).  usage: timeline.py <trace dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_fused" in r[2]]
for s0, s1 in list(zip(idx, idx[1:]))[-6:-4]:
    step = rows[s0:s1]; t0 = step[0][0]; prev = t0
    print("--- period %.1f us" % ((rows[s1][0] - t0) / 1e3))
    for st, en, name in step:
        short = name.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:50]
        print("  +%7.1f gap %6.1f dur %7.1f %s" % ((st - t0) / 1e3, (st - prev) / 1e3, (en - st) / 1e3, short)); prev = max(prev, en)
    print("  idle before next: %.1f" % ((rows[s1][0] - prev) / 1e3))
