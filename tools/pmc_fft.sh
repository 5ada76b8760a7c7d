#!/bin/bash
# SQ counters of the FFT passes (bench default command, 2 passes of counters)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcfft
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/pmcfft/a -- $CMD > $R/gpurun_out/pmcfft/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmcfft/b -- $CMD > $R/gpurun_out/pmcfft/b.log 2>&1
for d in a b; do python $R/tools/rocprof_summary.py --pmc $R/gpurun_out/pmcfft/$d | grep -A9 -E "k_fft_strided_h<10, 256, 2|k_fft_strided_h<10, 256, 0|k_fft_z_forward<10|k_deposit_cic_pull<16, false>"; done > $R/gpurun_out/pmcfft/summary.txt
cat $R/gpurun_out/pmcfft/summary.txt
rm -rf $R/gpurun_out/pmcfft/a $R/gpurun_out/pmcfft/b
