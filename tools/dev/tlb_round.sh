#!/bin/bash
# dev, ON THE GPU BOX: address-translation counters of the streaming kernel for a list of shapes.  usage: tlb_round.sh <tag> shape...
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for shape in "$@"; do
  i=0
  for ctrs in \
    "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
    "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum" \
    "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_LFIFO_FULL_sum GRBM_UTCL2_BUSY" \
    "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" ; do
    i=$((i+1))
    SHAPE=$shape rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/t_${shape//,/x}_$i -- python3 $R/tools/dev/fused_time.py > $O/t_${shape//,/x}_$i.log 2>&1
  done
done
python3 - <<PY | tee $O/tlb_summary.txt
import csv, glob, collections, os
print("build:", open("$R/profiles/BUILD_ID").read().strip() if os.path.exists("$R/profiles/BUILD_ID") else "?")
for d in sorted(set(p.rsplit("_", 1)[0] for p in glob.glob("$O/t_*_[0-9]"))):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "_[0-9]/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fused" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== k_fused,", os.path.basename(d)[2:], "(mean of the last 3 launches)")
    for c, v in sorted(acc.items()):
        v = v[-3:]
        print(f"  {c:48s} {sum(v) / len(v):16.0f}")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
