#!/bin/bash
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/fetch -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-times --no-train-step > $R/gpurun_out/$TAG.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/write -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-times --no-train-step > $R/gpurun_out/$TAG.write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-kernel-times --no-train-step > $R/gpurun_out/$TAG.trace.log 2>&1
tail -2 $R/gpurun_out/$TAG.fetch.log
