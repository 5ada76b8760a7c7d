#!/bin/bash
# usage: tools/pmc_sr.sh [variant.so] : SQ counters of the short-range tile sweep (k_sr_sweep_mfma), 256^3 / 512^3 uniform
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcsr
[ -n "$1" ] && export CONCEPT_GPU_LIB=$R/$1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcsr/p$i -- python $R/tools/sr_mfma_time.py ${SR_DIST:-uniform} > $R/gpurun_out/pmcsr/p$i.log 2>&1
  python $R/tools/rocprof_summary.py --pmc $R/gpurun_out/pmcsr/p$i | grep -A9 "k_sr_sweep_mfma"
done
python $R/tools/rocprof_summary.py $R/gpurun_out/pmcsr/p1 | grep "k_sr_sweep_mfma\|k_srm"
rm -rf $R/gpurun_out/pmcsr/p1 $R/gpurun_out/pmcsr/p2
