#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
python tools/dev/chunk_pre_diag.py > $O/chunk_pre_diag.txt 2>&1; tail -20 $O/chunk_pre_diag.txt
for i in 1 2 3; do
  for L in base ps65 wl4 ps65wl4; do echo -n "$L: "; P3D_CAPI_LIB=$R/build_dev/$L.so python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*F [0-9]* //'; done
done | tee $O/ab_fused.txt
cd /tmp && export TMPDIR=/tmp
for fv in 2 1; do
i=0
for ctrs in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_INT32" ; do
  i=$((i+1))
  P3D_FACES_V=$fv rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/v${fv}sq$i -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/v${fv}sq$i.log 2>&1
done
python3 - <<PY > $O/sq_summary_faces_v$fv.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/v${fv}sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = None
        for o in ("k_fused", "k_face_count_walk", "k_faces"):
            if o in n:
                k = o
                break
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print("==", k, "(c3, P3D_FACES_V=$fv)")
    for c, v in sorted(d.items()):
        v = v[-3:]
        print(f"  {c:28s} {sum(v)/len(v):16.0f}")
PY
done
paste $O/sq_summary_faces_v2.txt $O/sq_summary_faces_v1.txt | grep -A30 "k_faces"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
