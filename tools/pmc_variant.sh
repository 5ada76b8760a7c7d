#!/bin/bash
# usage: tools/pmc_variant.sh <variant.so> : SQ wave-cycle counters of the fused pass for a variant library
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcv
CONCEPT_GPU_LIB=$R/$1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/pmcv/a -- python $R/tools/fused_probe.py > $R/gpurun_out/pmcv/a.log 2>&1
python $R/tools/rocprof_summary.py --pmc $R/gpurun_out/pmcv/a | grep -A9 "k_gather_kick_tiled<2, 16, 2>"
python $R/tools/rocprof_summary.py $R/gpurun_out/pmcv/a | grep "k_gather_kick_tiled<2, 16, 2>"
rm -rf $R/gpurun_out/pmcv/a
