#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): L2 / fabric counter passes over bench.py's call stream, summarised per launch for
# every kernel of the call -- where the streaming kernel's read over-fetch goes (requests into L2, hits, misses, the
# 32 / 64 / 128-byte reads that leave L2 for the fabric).
# usage: tools/l2_round.sh <tag> [config]     -> gpurun_out/<tag>/l2_summary[_<config>].txt
# (program directly after `--`; --pmc passes carry --kernel-trace only; FETCH_SIZE never shares a pass)
tag=${1:-r04}; cfg=${2:-c3}; sfx=""; [ "$cfg" != c3 ] && sfx="_$cfg"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_ATOMIC_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum" \
  "TCC_STREAMING_REQ_sum TCC_NC_REQ_sum TCC_UC_REQ_sum TCC_CC_REQ_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/l2$sfx$i -- python3 $R/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-modes --no-live-traffic > $O/l2$sfx$i.log 2>&1
done
python3 - <<PY > $O/l2_summary$sfx.txt
import csv, glob, collections, subprocess
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/l2$sfx[0-9]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = None
        for o in ("k_fused", "k_face_count_walk", "k_faces", "k_stack_finish", "k_chunk_prefix"):
            if o in n:
                k = o
                break
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
import os
head = open("$R/profiles/BUILD_ID").read().strip() if os.path.exists("$R/profiles/BUILD_ID") else \
    "lib=" + open("$R/primitive3d_amd/libp3dmc.so.stamp").read().strip()[:16]
print("build:", head, " config: $cfg  (per launch, mean of the last 3 launches of each pass)")
for k, d in sorted(acc.items()):
    print("==", k)
    m = {}
    for c, v in sorted(d.items()):
        v = v[-3:]
        m[c] = sum(v) / len(v)
        print(f"  {c:30s} {m[c]:16.0f}")
    if "TCC_EA0_RDREQ_32B_sum" in m and "TCC_EA0_RDREQ_128B_sum" in m:
        b = 32 * m["TCC_EA0_RDREQ_32B_sum"] + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * m["TCC_EA0_RDREQ_128B_sum"]
        print(f"  -> bytes read from the fabric by request size  {b / 1e6:10.1f} MB")
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
        print(f"  -> L2 hit rate  {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
PY
cat $O/l2_summary$sfx.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
