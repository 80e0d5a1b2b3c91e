R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv.py tests/test_gpu_network.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
for i in 1 2; do
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dump-profile $O/prof.json) > $O/bench$i.log 2> $O/bench$i.err; echo "bench rc=$?" | tee -a $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench$i.log') if q.startswith('{')][-1]; d=json.loads(l)
print('fps %.1f ms %.3f dom %.4f frac %.3f step %.3f | single %.3f ms (dom %.4f ms frac %.3f) | precise %.2f ms batch8 %.2f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['step_roofline']['frac'],d['single_image']['ms_per_call'],d['single_image']['roofline_dominant_kernel']['avg_launch_ms'],d['single_image']['roofline_dominant_kernel']['frac'],d['precise']['ms_per_image'],d['precise']['batch8']['ms_per_image']))
PY
done
tail -2 $O/pytest.log
python - <<'PY'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04g'
t=[e for e in json.load(open(O+'/prof.json'))['entries'] if e['kernel'].endswith(':units')]
print('tail units ms:', round(sum(e['total_ms'] for e in t),4), [round(e['total_ms'],4) for e in t[:8]])
PY
