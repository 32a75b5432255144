#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench lines of every single-GPU config + rocprofv3 kernel stats + HBM traffic
# counters + SQ counters of the headline config.   usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>/
tag=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err   # (carries other_configs: c2, c5, c4 on one GPU)
python bench.py --config c2 --steps 50 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --config c5 --steps 10 --warmup 3 > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4_1gpu.json 2> $O/bench_c4.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-modes --no-live-traffic > $O/stats.log 2>&1
# modes.fresh_grid's call stream alone: four distinct 512^3 grids in turn (the headline re-extracts ONE resident grid)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fresh -- python3 $R/bench.py --child fresh --steps 40 --warmup 8 > $O/stats_fresh.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 $R/bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic > $O/stats_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 $R/bench.py --config c2 --steps 50 --warmup 5 --no-cpu-baseline --no-live-traffic > $O/stats_c2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-modes --no-live-traffic > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-modes --no-live-traffic > $O/pmc_write.log 2>&1
# calibration of the counters on a kernel with a KNOWN byte count and the same access shape:
# the staged extractor's k_classify reads the 512 MiB field once with dword loads and writes 16 MiB of sign words
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -- python3 $R/tools/calibrate_fetch.py > $O/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -- python3 $R/tools/calibrate_fetch.py > $O/cal_write.log 2>&1
cd $R && python3 tools/summarize_profile.py $O > $O/summary.txt 2>&1
python3 - >> $O/summary.txt <<PY
import csv
def avg(path, name):
    for r in csv.DictReader(open(path)):
        if name in r["Name"]:
            return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    return None
try:
    a, b = avg("$O/kernel_stats_ours.csv", "k_fused"), avg("$O/kernel_stats_fresh_ours.csv", "k_fused")
    print("k_fused, rocprofv3 average: one resident grid re-extracted %.1f us (%d calls); four distinct grids in turn %.1f us (%d calls)"
          " -> %.3f / %.3f of the 8 TB/s peak on the algorithmic 536 870 912 B" % (a[0], a[1], b[0], b[1], 536870912 / a[0] / 8e6, 536870912 / b[0] / 8e6))
except Exception as e:
    print("fresh-grid comparison unavailable:", e)
PY
bash tools/pmc_round.sh $tag > /dev/null 2>&1
bash tools/pmc_round.sh $tag c5 > /dev/null 2>&1
cat $O/summary.txt $O/sq_summary.txt $O/sq_summary_c5.txt
# keep what travels back small: the raw traces are summarised above
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
