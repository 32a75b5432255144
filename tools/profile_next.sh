#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench lines + rocprofv3 kernel stats of the section-8(f4) rows (tools/bench_next.py).
# usage: tools/profile_next.sh <tag>   -> gpurun_out/<tag>/
tag=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python tools/bench_next.py > $O/bench_next.json 2> $O/bench_next.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tetra -- python3 $R/tools/bench_next.py --what tetra --order lattice --steps 5 > $O/stats_tetra.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_rc -- python3 $R/tools/bench_next.py --what raycast --steps 5 > $O/stats_rc.log 2>&1
for w in tetra rc; do f=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$w.csv; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/bench_next.json; head -25 $O/kernel_stats_tetra.csv | cut -c1-160; head -8 $O/kernel_stats_rc.csv | cut -c1-160
