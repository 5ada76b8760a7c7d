#!/bin/bash
# HBM-side traffic of tools/xcd_handover_probe's kernels (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE
# in separate passes): does the hand-over's scratch window stay in the XCD's L2?
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
MODE=${1:-1}
for rounds in 8 4 2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/ho_$c
    rocprofv3 --kernel-trace --pmc $c -d /tmp/ho_$c -- $R/tools/xcd_handover_probe 1024 $rounds $MODE > /tmp/ho_$c.log 2>&1
    echo "== rounds $rounds, $c (mean per launch, raw counter: see MI355X_MICROARCH.md for its unit)"
    python $R/tools/rocprof_summary.py --pmc /tmp/ho_$c | grep -A2 "k_rows\|k_cols\|k_handover" | grep -v "^--"
  done
done
