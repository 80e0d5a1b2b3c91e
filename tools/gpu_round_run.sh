# Round measurement on the GPU box: GPU tests, rocprofv3 kernel stats, PMC passes (one counter group per pass, as
# MI355X_MICROARCH.md prescribes), their summary, then bench.py (whose roofline.traffic reads that summary).
# Usage: gpurun -- 'bash tools/gpu_round_run.sh r02'; afterwards here: python tools/summarize_profiles.py r02 r02
ROUND=${1:-r01}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/$ROUND; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60) > $O/pytest_gpu.log
(timeout 300 python tools/profile_driver.py --batch 32 --steps 3 --gen 5 --profile-json $O/prof_gen5.json) > $O/drv_old.log 2>&1
(timeout 300 python tools/profile_driver.py --batch 32 --steps 3 --profile-json $O/prof_strip_tiles.json) > $O/drv_new.log 2>&1
cd /tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/rp_stats -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 3) > $O/rp_stats.log 2>&1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $P | cut -d" " -f1)
  (timeout 300 rocprofv3 --pmc $P --kernel-trace -d $O/pmc_$N -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 1) > $O/pmc_$N.log 2>&1
done
cd $R
python tools/summarize_profiles.py $ROUND $ROUND > $O/summary.log 2>&1
(timeout 600 python bench.py --steps 5 --warmup 2 --dump-profile $O/prof_bench.json) > $O/bench.log 2>&1
# the same command under rocprofv3 (kernel durations must agree with the HIP-event figures in the bench line)
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $O/rp_bench -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline) > $O/rp_bench.log 2>&1
# single-image path (BASELINE config 2): kernel trace at batch 1 (gaps between launches = span - sum of durations)
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/rp_b1 -o b1 --output-format csv -- python $R/tools/profile_driver.py --batch 1 --steps 20) > $O/rp_b1.log 2>&1
cd $R
(timeout 300 python tools/power_probe.py) > $O/power_probe.txt 2>&1
(timeout 400 python tools/splitk_tune.py --batches 1 2 3 4 6) > $O/splitk_tune.txt 2>&1
(timeout 300 python tools/wino_check.py --json $O/wino_check.json) > $O/wino_check.txt 2>&1
(timeout 300 python tools/wino_batch_sweep.py) > $O/wino_batch_sweep.txt 2>&1
(timeout 300 python tools/soak.py --steps 300) > $O/soak.txt 2>&1
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')") > $O/smoke.log 2>&1
tail -1 $O/smoke.log; tail -3 $O/pytest_gpu.log; tail -2 $O/drv_old.log $O/drv_new.log; tail -1 $O/bench.log; du -sh $O
