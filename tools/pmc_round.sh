#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): SQ counter passes over bench.py's call stream for every kernel of the call.
# usage: tools/pmc_round.sh <tag> [config]     -> gpurun_out/<tag>/sq_summary[_<config>].txt   (config: c3 (default), c2, c5)
# (program directly after `--`; --pmc passes carry --kernel-trace only)
tag=${1:-r03}; cfg=${2:-c3}; sfx=""; [ "$cfg" != c3 ] && sfx="_$cfg"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_INT32" ; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/sq$sfx$i -- python3 $R/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-modes --no-live-traffic > $O/sq$sfx$i.log 2>&1
done
python3 - <<PY > $O/sq_summary$sfx.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/sq$sfx[0-9]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = None
        for o in ("k_fused", "k_face_count_walk", "k_faces", "k_mesh", "k_face_total"):
            if o in n:
                k = o
                break
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
import os
print("build:", open("$R/profiles/BUILD_ID").read().strip() if os.path.exists("$R/profiles/BUILD_ID") else "unknown")
for k, d in sorted(acc.items()):
    print("==", k, "($cfg)")
    for c, v in sorted(d.items()):
        v = v[-3:]
        print(f"  {c:28s} {sum(v)/len(v):16.0f}")
PY
cat $O/sq_summary$sfx.txt
