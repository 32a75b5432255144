"""dev: every kernel of the LAST call in a rocprofv3 kernel trace, split at a marker kernel.
usage: timeline_all.py <trace dir> <marker kernel substring>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[2]]
s0, s1 = idx[-3], idx[-2]
step = rows[s0:s1]; t0 = step[0][0]; prev = t0
print("--- period %.1f us, %d kernels" % ((rows[s1][0] - t0) / 1e3, len(step)))
for st, en, name in step:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rp::").replace("rocprim::ROCPRIM_400001_NS::detail::", "rp1::")[:60]
    print("  +%7.1f gap %6.1f dur %7.1f %s" % ((st - t0) / 1e3, (st - prev) / 1e3, (en - st) / 1e3, short)); prev = max(prev, en)
