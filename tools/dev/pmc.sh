#!/bin/bash
# usage: tools/dev/pmc.sh "<counters>" <outname> -- python3 script...   (one --pmc pass)
ctr="$1"; out="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/$out; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- "$@" > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
python3 - <<PY
import csv, glob, collections
fs = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$out/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "k_fused" in k or "k_faces" in k or "k_classify" in k:
        print(k, {c: round(sum(v[-3:]) / len(v[-3:])) for c, v in d.items()})
PY
