# Round-5 refresh after the detect_precise / pp_limbs changes: GPU suite, smoke, bench line (full), rocprofv3 stats of the bench command and of detect_precise
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee $O/summary.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.log
(timeout 900 python bench.py --steps 20 --warmup 3 --dump-profile $O/prof_bench.json) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.log
cd /tmp; rm -rf $O/rp_bench $O/rp_precise
(timeout 600 rocprofv3 --kernel-trace --stats -d $O/rp_bench -o bench --output-format csv -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras) > $O/rp_bench.log 2>&1
(timeout 400 rocprofv3 --kernel-trace --stats -d $O/rp_precise -o precise --output-format csv -- python $R/tools/precise_bench_driver.py) > $O/rp_precise.log 2>&1
cd $R; rm -f $O/rp_*/*trace.csv
tail -3 $O/pytest_gpu.log; cat $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench.log') if q.startswith('{')][-1]; d=json.loads(l)
print('fps %.1f ms %.3f dom %.4f frac %.3f step %.3f | single %.3f ms | precise %.2f ms batch8 %.2f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['step_roofline']['frac'],d['single_image']['ms_per_call'],d['precise']['ms_per_image'],d['precise']['batch8']['ms_per_image']))
print(json.dumps(d['precise'].get('keypoint_match_vs_precise_ref'))[:500])
PY
