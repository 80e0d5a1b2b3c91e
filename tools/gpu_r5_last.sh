# Round-5 closing run on the final tree: whole GPU suite, smoke, the bench line (default flags)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05last; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.log
tail -3 $O/pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.log
(timeout 600 python bench.py) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench.log') if q.startswith('{')][-1]; d=json.loads(l)
print('fps %.1f ms %.3f dom %.4f frac %.3f step %.3f | single %.3f ms | precise %.2f ms batch8 %.2f | cpu %.2f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['step_roofline']['frac'],d['single_image']['ms_per_call'],d['precise']['ms_per_image'],d['precise']['batch8']['ms_per_image'],d['cpu_baseline']['value']))
PY
