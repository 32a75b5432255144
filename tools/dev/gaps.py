"""dev: per-step GPU timeline from a rocprofv3 kernel trace of bench.py (gaps between kernels)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
# keep the last 8 steps: a step starts at k_fused
idx = [i for i, r in enumerate(rows) if "k_fused" in r[2]]
for s0, s1 in list(zip(idx, idx[1:]))[-4:]:
    step = rows[s0:s1]
    t0 = step[0][0]
    print("--- step, period %.1f us" % ((rows[s1][0] - t0) / 1e3))
    prev_end = t0
    for st, en, name in step:
        short = name.split("(")[0].split("::")[-1][:40]
        print("  +%7.1f  gap %6.1f  dur %7.1f  %s" % ((st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3, short))
        prev_end = max(prev_end, en)
    print("  idle before next k_fused: %.1f us" % ((rows[s1][0] - prev_end) / 1e3))
