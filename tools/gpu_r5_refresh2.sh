# Round-5 last refresh (clustered transform adopted): GPU suite, smoke, bench line, force-gather line, rocprofv3 stats of the bench command / driver / batch 1 /
# landscape / detect_precise, the five PMC passes
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee $O/summary.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.log
(timeout 900 python bench.py --steps 20 --warmup 3 --dump-profile $O/prof_bench.json) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.log
(timeout 300 python bench.py --steps 10 --warmup 2 --force-gather --no-cpu-baseline --no-extras) > $O/bench_force_gather.log 2> $O/bench_force_gather.err; echo "bench force-gather rc=$?" | tee -a $O/summary.log
cd /tmp; rm -rf $O/rp_bench $O/rp_precise $O/rp_stats $O/rp_b1 $O/rp_rect $O/pmc_*
(timeout 600 rocprofv3 --kernel-trace --stats -d $O/rp_bench -o bench --output-format csv -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras) > $O/rp_bench.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/rp_stats -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 3) > $O/rp_stats.log 2>&1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $P | cut -d" " -f1)
  (timeout 300 rocprofv3 --pmc $P --kernel-trace -d $O/pmc_$N -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 1) > $O/pmc_$N.log 2>&1
done
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/rp_b1 -o b1 --output-format csv -- python $R/tools/profile_driver.py --batch 1 --steps 20) > $O/rp_b1.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/rp_rect -o rect --output-format csv -- python $R/tools/rect_time.py --h 368 --w 496 --batch 32 --steps 3) > $O/rp_rect.log 2>&1
(timeout 400 rocprofv3 --kernel-trace --stats -d $O/rp_precise -o precise --output-format csv -- python $R/tools/precise_bench_driver.py) > $O/rp_precise.log 2>&1
cd $R; rm -f $O/rp_*/*trace.csv $O/pmc_*/*agent_info.csv
tail -3 $O/pytest_gpu.log; cat $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench.log') if q.startswith('{')][-1]; d=json.loads(l)
print('fps %.1f ms %.3f dom %.4f frac %.3f step %.3f | single %.3f ms | precise %.2f ms batch8 %.2f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['step_roofline']['frac'],d['single_image']['ms_per_call'],d['precise']['ms_per_image'],d['precise']['batch8']['ms_per_image']))
PY
