#!/bin/bash
# SQ counters of the fused kick + drift + scatter pass (tools/fused_probe.py), two passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcf
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/pmcf/a -- python $R/tools/fused_probe.py > $R/gpurun_out/pmcf/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES -d $R/gpurun_out/pmcf/b -- python $R/tools/fused_probe.py > $R/gpurun_out/pmcf/b.log 2>&1
python $R/tools/rocprof_summary.py --pmc $R/gpurun_out/pmcf/a | grep -A9 "k_gather_kick_tiled<2, 16, 2>" > $R/gpurun_out/pmcf/summary.txt
python $R/tools/rocprof_summary.py --pmc $R/gpurun_out/pmcf/b | grep -A9 "k_gather_kick_tiled<2, 16, 2>" >> $R/gpurun_out/pmcf/summary.txt
cat $R/gpurun_out/pmcf/summary.txt; tail -3 $R/gpurun_out/pmcf/a.log
rm -rf $R/gpurun_out/pmcf/a $R/gpurun_out/pmcf/b
