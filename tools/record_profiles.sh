#!/bin/bash
# Records the round's measurements on the GPU box into gpurun_out/profiles_new/ (copied into
# profiles/ afterwards):  tools/record_profiles.sh r05 <commit>
# One-liners a driver can reproduce are listed in profiles/README.md.
set -u
TAG=${1:-r06}; COMMIT=${2:-unknown}; MODE=${3:-full}   # quick: kernel statistics, HBM traffic and the default line only
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_new; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && python -m concept_amd.build > /dev/null 2>&1)  # (a library older than csrc/ would be profiled under the wrong hash)
CMD="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-configs"
# 1. per-kernel statistics of the default command
rocprofv3 --kernel-trace --stats -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
python $R/tools/rocprof_summary.py $OUT/stats $OUT/${TAG}_rocprof_kernel_stats_ns.txt > /dev/null
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_f -- $CMD > $OUT/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_w -- $CMD > $OUT/pmc_w.log 2>&1
python $R/tools/rocprof_summary.py --pmc $OUT/pmc_f > $OUT/${TAG}_pmc_FETCH_SIZE_per_kernel.txt
python $R/tools/rocprof_summary.py --pmc $OUT/pmc_w > $OUT/${TAG}_pmc_WRITE_SIZE_per_kernel.txt
python $R/tools/pmc_traffic.py $OUT/pmc_f $OUT/pmc_w $OUT/${TAG}_pmc_hbm_traffic.json ns_256M_1024 $COMMIT "python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-configs" > $OUT/pmc_traffic.log 2>&1
cp $OUT/${TAG}_pmc_hbm_traffic.json $R/profiles/${TAG}_pmc_hbm_traffic.json   # (so that the lines below quote it)
# 3. P3M: statistics and SQ counters of the sweep
if [ "$MODE" = quick ]; then
  (cd $R && python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_ns_full_default.json)
  rm -rf $OUT/stats $OUT/pmc_f $OUT/pmc_w; ls -la $OUT; exit 0
fi
P3M="python $R/bench.py --workload c2_256c_512 --p3m --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats_p3m -- $P3M > $OUT/stats_p3m.log 2>&1
python $R/tools/rocprof_summary.py $OUT/stats_p3m $OUT/${TAG}_rocprof_kernel_stats_p3m.txt > /dev/null
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc_sr -- $P3M > $OUT/pmc_sr.log 2>&1
python $R/tools/rocprof_summary.py --pmc $OUT/pmc_sr > $OUT/${TAG}_pmc_sr_sweep.txt
# 4. the bench lines
cd $R
python bench.py --workload c2_256c_512 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c2_256c_512.json
python bench.py --workload c2_256c_512 --p3m --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_p3m_c3_256c_512.json
python bench.py --workload c2_256c_512 --p3m --dist clustered --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_p3m_clustered.json
python bench.py --dist clustered --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_ns_clustered.json
python bench.py --no-fused --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_ns_unfused.json
python bench.py --workload c3_1024c_2048 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c3_1024c_2048.json
CONCEPT_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_ns_forced_dist_1rank_rccl.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_ns_full_default.json
python bench.py --workload ns_256M_1024 --p3m --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_p3m_ns_256M_1024.json
python bench.py --weak --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_weak_1gpu.json
python bench.py --workload c2_256c_512 --gpus 8 --dry-links --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c2_8ranks_gloo_dry_links.json
# 6. the dense tiles' sweep (round 4): against the cells sweep by itself, its counters, its phases
python tools/sr_dense_check.py scan big time 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_sr_dense_vs_cells.txt
(SR_DIST=clustered $R/tools/pmc_srd.sh 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_pmc_sr_dense.txt)
python tools/sr_dense_time.py clustered 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_sr_dense_phases.txt
CONCEPT_GPU_SR_DENSE_MIN=0 python tools/sr_dense_time.py clustered 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_sr_dense_phases.txt
# 7. configs[4]'s shape on one GPU (round 5): the line and its kernel statistics
python bench.py --workload c4_nonlinnu_1gpu --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/${TAG}_bench_c4_nonlinnu_1gpu.json
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats_c4 -- python $R/bench.py --workload c4_nonlinnu_1gpu --steps 3 --warmup 1 > $OUT/stats_c4.log 2>&1)
python tools/rocprof_summary.py $OUT/stats_c4 $OUT/${TAG}_rocprof_kernel_stats_c4.txt > /dev/null; rm -rf $OUT/stats_c4
# 8. the rung loop (round 6): kernel statistics of the P3M time loop with 8 rungs, what a sub-step's
# sweep and list cost by lowest active rung, the bench's leg for both boxes
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats_soak -- python $R/tools/soak_p3m.py 0.04 > $OUT/stats_soak.log 2>&1)
(echo "# rocprofv3 --kernel-trace --stats -- python tools/soak_p3m.py 0.04  (the P3M time loop with 8 rungs, 256^3 / 512^3, a = 0.02 -> 0.04)"; grep 'base steps' $OUT/stats_soak.log; ROCPROF_GAPS=1 python tools/rocprof_summary.py $OUT/stats_soak) > $OUT/${TAG}_rocprof_kernel_stats_soak_p3m.txt; rm -rf $OUT/stats_soak
python tools/sr_rung_cost.py uniform 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_sr_rung_cost.txt
python bench.py --rung-loop uniform --steps 12 2>/dev/null | tail -1 > $OUT/${TAG}_bench_rung_loop_uniform.json
python bench.py --rung-loop clustered --steps 10 2>/dev/null | tail -1 > $OUT/${TAG}_bench_rung_loop_clustered.json
(python tools/soak_p3m.py 0.04 2>&1 | grep 'base steps'; SOAK_DIST=clustered python tools/soak_p3m.py 0.025 2>&1 | grep 'base steps') > $OUT/${TAG}_soak_p3m.txt
(SR_DIST=uniform $R/tools/pmc_srd.sh 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_pmc_sr_blocks.txt)
# 9. the 1-rank values an N-rank run of the driver's command compares its sample with
cp .bench_verify/ns_256M_1024_seed1_thermal0.2_steps25.json $OUT/${TAG}_bench_verify_ns_256M_1024_seed1_thermal0.2_steps25.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extra-configs > /dev/null 2>&1
cp .bench_verify/ns_256M_1024_seed1_thermal0.2_steps7.json $OUT/${TAG}_bench_verify_ns_256M_1024_seed1_thermal0.2_steps7.json 2>/dev/null
# heavy tiles first (cgk_tile_order) against the plain walk, one process
python tools/fused_order_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_tile_order_ab.txt
rm -rf $OUT/stats $OUT/pmc_f $OUT/pmc_w $OUT/stats_p3m $OUT/pmc_sr
ls -la $OUT
