#!/bin/bash
# L2 (TCC) counters of the fused kick + drift + scatter pass (tools/fused_probe.py): hit rate and
# the requests that go to the fabric, in two passes (TCC has 4 counters per pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcl2
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $R/gpurun_out/pmcl2/a -- python $R/tools/fused_probe.py > $R/gpurun_out/pmcl2/a.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $R/gpurun_out/pmcl2/b -- python $R/tools/fused_probe.py > $R/gpurun_out/pmcl2/b.log 2>&1
python $R/tools/rocprof_summary.py --pmc $R/gpurun_out/pmcl2/a | grep -A5 "k_gather_kick_tiled<2, 16, 2>" > $R/gpurun_out/pmcl2/summary.txt
python $R/tools/rocprof_summary.py --pmc $R/gpurun_out/pmcl2/b | grep -A5 "k_gather_kick_tiled<2, 16, 2>" >> $R/gpurun_out/pmcl2/summary.txt
cat $R/gpurun_out/pmcl2/summary.txt; tail -3 $R/gpurun_out/pmcl2/a.log; tail -3 $R/gpurun_out/pmcl2/b.log
rm -rf $R/gpurun_out/pmcl2/a $R/gpurun_out/pmcl2/b
