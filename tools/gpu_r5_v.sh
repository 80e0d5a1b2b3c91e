# Round-5 GPU call V: clustered transform adopted (slot 20): bit-exactness suites, bench x2
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05v; mkdir -p $O; cd $R
(timeout 1200 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv.py tests/test_gpu_network.py tests/test_gpu_reference_goldens.py tests/test_gpu_properties.py tests/test_gpu_selection.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -4 $O/pytest.log
for i in 1 2; do
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --dump-profile $O/prof.json) > $O/bench$i.log 2> $O/bench$i.err; echo "bench rc=$?" | tee -a $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench$i.log') if q.startswith('{')][-1]; d=json.loads(l)
print('fps %.1f ms %.3f dom %.4f ms frac %.3f step frac %.3f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['step_roofline']['frac']))
PY
done
