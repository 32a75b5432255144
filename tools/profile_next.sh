#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench lines, rocprofv3 kernel stats and PMC passes (HBM bytes, SQ counters) of the
# section-8(f4) rows (tools/bench_next.py).   usage: tools/profile_next.sh <tag>   -> gpurun_out/<tag>/
# (program directly after `--`; --pmc passes carry --kernel-trace only)
tag=${1:-next}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python tools/bench_next.py > $O/bench_next.json 2> $O/bench_next.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tetra -- python3 $R/tools/bench_next.py --what tetra --order lattice --steps 5 > $O/stats_tetra.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_rc -- python3 $R/tools/bench_next.py --what raycast --steps 5 > $O/stats_rc.log 2>&1
for w in tetra rc; do f=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$w.csv; done
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM" ; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/pmc_tetra$i -- python3 $R/tools/bench_next.py --what tetra --order lattice --steps 3 > $O/pmc_tetra$i.log 2>&1
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/pmc_rc$i -- python3 $R/tools/bench_next.py --what raycast --steps 3 > $O/pmc_rc$i.log 2>&1
done
python3 - <<PY > $O/next_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = next((o for o in ("k_mt_classify", "k_mt_edges", "k_mt_faces", "k_mt_vertices", "k_mt_ranks", "k_mt_occ", "k_raycast") if o in n), None)
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("PMC counters per launch (mean of the last 3 launches); FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them")
print("(profiles/r02/summary.txt: the read counter needs a factor 1.999 for dword-per-lane streams on gfx950; gathers are not calibrated)")
for k, d in sorted(acc.items()):
    print("==", k)
    for c, v in sorted(d.items()):
        v = v[-3:]
        print(f"  {c:28s} {sum(v)/len(v):16.0f}")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
cat $O/bench_next.json | cut -c1-300; cat $O/next_pmc.txt
