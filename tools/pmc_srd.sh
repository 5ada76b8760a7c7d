#!/bin/bash
# usage: tools/pmc_srd.sh [variant.so] : kernel times and SQ counters of the short-range sweep with the dense tiles'
# kernel (k_sr_sweep_dense beside k_sr_sweep_blocks / k_sr_sweep_cells), 256^3 / 512^3, SR_DIST=clustered (default) or uniform
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcsrd
[ -n "$1" ] && export CONCEPT_GPU_LIB=$R/$1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmcsrd/s -- python $R/tools/sr_dense_time.py ${SR_DIST:-clustered} > $R/gpurun_out/pmcsrd/s.log 2>&1
python $R/tools/rocprof_summary.py $R/gpurun_out/pmcsrd/s | grep "k_sr\|Name\|name" | cut -c1-200
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcsrd/p$i -- python $R/tools/sr_dense_time.py ${SR_DIST:-clustered} > $R/gpurun_out/pmcsrd/p$i.log 2>&1
  python $R/tools/rocprof_summary.py --pmc $R/gpurun_out/pmcsrd/p$i | grep -A9 "k_sr_sweep_dense\|k_sr_sweep_blocks" | cut -c1-160
done
rm -rf $R/gpurun_out/pmcsrd/s $R/gpurun_out/pmcsrd/p1 $R/gpurun_out/pmcsrd/p2
