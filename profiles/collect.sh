#!/bin/bash
# usage (on the GPU box, from the repo root):  bash profiles/collect.sh <tag>
# Collects, for the serial eager schedule of bench.py's default workload (cfg3 view mode; kernels of concurrent views
# would share the GPU and inflate each other's durations):
#   1. rocprofv3 --kernel-trace --stats                        -> per-kernel durations
#   2. three SQ counter passes (--pmc with --kernel-trace only) -> instruction mix / waits / LDS
#   3. FETCH_SIZE and WRITE_SIZE in separate passes            -> HBM traffic per launch
# Raw output goes to gpurun_out/<tag>/; profiles/summarise.py turns it into the committed summaries.
# Every pass runs under `timeout`: a counter set the hardware cannot collect makes rocprofv3 abort and then hang (round 4:
# eight TCC counters in one pass cost 25 minutes of box time); keep passes at <= 8 SQ or <= 3 TCC counters.
TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
# serial eager schedule of the headline body (fused direct entry points, one view at a time, no graph)
B="python $R/bench.py --streams 1 --views-per-step 1 --no-graph --no-cpu-baseline --no-kernel-times --no-train-step --min-seconds 0"
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 16 --warmup 2 > $O/trace.log 2>&1
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc1 -- $B --steps 4 --warmup 1 > $O/pmc1.log 2>&1
timeout 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc2 -- $B --steps 4 --warmup 1 > $O/pmc2.log 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $O/pmc3 -- $B --steps 4 --warmup 1 > $O/pmc3.log 2>&1
# round 6: the clock the kernels really ran at (GRBM_GUI_ACTIVE / duration; MI355X_MICROARCH.md, "DVFS give-back") and the

timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc4 -- $B --steps 4 --warmup 1 > $O/pmc4.log 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B --steps 4 --warmup 1 > $O/fetch.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B --steps 4 --warmup 1 > $O/write.log 2>&1
# the operator-API instances (GaussianRasterizer: gated unit backward, general training / colour / all_map / all-gradient backward)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_general -- python $R/profiles/probes/general_instances.py cfg3 12 > $O/trace_general.log 2>&1
# the graph-replay schedule of the default bench command, kernel trace only (durations overlap across streams)
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_graph -- python $R/bench.py --no-cpu-baseline --no-kernel-times --no-train-step --steps 4 --warmup 1 --min-seconds 0 > $O/trace_graph.log 2>&1
find $O -name "*.csv" | wc -l
