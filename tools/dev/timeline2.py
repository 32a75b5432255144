"""dev: kernel timeline (start, duration, gap to the previous kernel's end) of the last N kernels of a rocprofv3 kernel trace.
usage: timeline2.py <kernel_trace.csv> [N]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -40:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"]
    for k in ("k_fused", "k_face_count_walk", "k_faces", "k_zero_words", "k_early_header", "k_export", "k_face_total", "k_chunk_prefix", "k_stack_finish"):
        if k in name: name = k; break
    print("%9.1f us  dur %7.1f  gap %7.1f  %s" % (s / 1e3, (e - s) / 1e3, ((s - prev_end) / 1e3) if prev_end is not None else 0.0, name[:70]))
    prev_end = e if prev_end is None else max(prev_end, e)
