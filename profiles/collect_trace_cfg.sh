#!/bin/bash
# usage (on the GPU box, from the repo root):  bash profiles/collect_trace_cfg.sh <tag> <cfgN> [<cfgM> ...]
# rocprofv3 --kernel-trace --stats of the serial eager schedule of bench.py's headline body (one view at a time, no graph)
# for the named configs -> gpurun_out/<tag>/trace_<cfg>/ ; profiles/summarise_trace_cfg.py turns them into
# profiles/<round>_kernel_stats_<cfg>.csv.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
for CFG in "$@"; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$CFG -- python $R/bench.py --config $CFG --streams 1 --views-per-step 1 --no-graph --no-cpu-baseline --no-kernel-times --no-train-step --min-seconds 0 --steps 16 --warmup 2 > $O/trace_$CFG.log 2>&1
done
find $O -name "*_kernel_stats.csv" | wc -l
