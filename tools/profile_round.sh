#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench line + rocprofv3 kernel stats + HBM traffic counters.
# usage: tools/profile_round.sh <tag>
tag=${1:-r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R && python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc_write.log 2>&1
# calibration of the counters on a kernel with a KNOWN byte count and the same access shape:
# the staged extractor's k_classify reads the 512 MiB field once with dword loads and writes 16 MiB of sign words
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -- python3 $R/tools/calibrate_fetch.py > $O/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -- python3 $R/tools/calibrate_fetch.py > $O/cal_write.log 2>&1
cd $R && python3 tools/summarize_profile.py $O > $O/summary.txt 2>&1; cat $O/summary.txt
