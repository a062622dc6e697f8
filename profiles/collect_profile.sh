#!/bin/bash
# usage: prof.sh <tag>  -- kernel trace + stats, then PMC passes
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-kernel-times > $R/gpurun_out/$TAG.trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/pmc1 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-times > $R/gpurun_out/$TAG.pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/pmc2 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-times > $R/gpurun_out/$TAG.pmc2.log 2>&1
find $R/gpurun_out/$TAG -name "*.csv" | head -20
