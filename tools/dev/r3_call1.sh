#!/bin/bash
# round 3, first GPU call: tests, bench, A/B of the two face kernels, fp16 SQ counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
for i in 1 2 3; do
  for v in 1 2; do echo -n "faces_v=$v: "; P3D_FACES_V=$v python tools/dev/fused_time.py 2>&1 | tail -1; done
done | tee $O/ab_faces.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
P3D_FACES_V=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_v1.json 2>> $O/bench.err; head -c 400 $O/bench_v1.json
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_INT32" ; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/c5sq$i -- python3 $R/bench.py --config c5 --steps 3 --warmup 2 --no-cpu-baseline > $O/c5sq$i.log 2>&1
done
python3 - <<PY > $O/sq_summary_c5.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/c5sq*/**/*counter_collection.csv", recursive=True):
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
    print("==", k, "(config c5: 32 x 256^3 fp16)")
    for c, v in sorted(d.items()):
        v = v[-3:]
        print(f"  {c:28s} {sum(v)/len(v):16.0f}")
PY
cat $O/sq_summary_c5.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
