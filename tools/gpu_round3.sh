# Round-3 measurement on the GPU box (no test suite: tools/gpu_full.sh runs that): per-layer profile, rocprofv3 kernel stats, PMC passes
# (one counter group per pass, as MI355X_MICROARCH.md prescribes), bench.py, the same under rocprofv3, batch-1 trace.
# Usage: gpurun -- 'bash tools/gpu_round3.sh r03'; afterwards here: python tools/summarize_profiles.py r03 r03
ROUND=${1:-r03}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/$ROUND; mkdir -p $O; cd /tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/rp_stats -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 3) > $O/rp_stats.log 2>&1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $P | cut -d" " -f1)
  (timeout 300 rocprofv3 --pmc $P --kernel-trace -d $O/pmc_$N -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 1) > $O/pmc_$N.log 2>&1
done
cd $R
(timeout 600 python bench.py --steps 5 --warmup 2 --dump-profile $O/prof_bench.json) > $O/bench.log 2> $O/bench.err
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $O/rp_bench -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline) > $O/rp_bench.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/rp_b1 -o b1 --output-format csv -- python $R/tools/profile_driver.py --batch 1 --steps 20) > $O/rp_b1.log 2>&1
cd $R
rm -f $O/rp_stats/*trace.csv $O/rp_bench/*trace.csv $O/rp_b1/*trace.csv
tail -c 600 $O/bench.log; du -sh $O
