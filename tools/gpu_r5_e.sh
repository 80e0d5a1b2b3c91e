# Round-5 GPU call E: the whole GPU suite + smoke under the new defaults (conv1_wino), bench with the extras
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.log
tail -25 $O/pytest_gpu.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.log
(timeout 900 python bench.py --steps 20 --warmup 3 --dump-profile $O/prof_bench.json) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench.log') if q.startswith('{')][-1]; d=json.loads(l)
print('fps %.1f ms %.3f dom %.4f frac %.3f step %.3f | single %.3f ms | precise %.2f ms batch8 %.2f | rect %s'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['step_roofline']['frac'],d['single_image']['ms_per_call'],d['precise']['ms_per_image'],d['precise']['batch8']['ms_per_image'], json.dumps(d['rect_368x496'])[:300]))
print(json.dumps(d['keypoint_match'])[:600])
print(json.dumps(d['precise'].get('keypoint_match_vs_precise_ref'))[:600])
print(json.dumps(d['cpu_baseline'])[:300])
PY
