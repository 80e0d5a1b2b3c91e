# Round-5 GPU call D: conv1_wino_kernel (conv1_1 + Winograd conv1_2 in one launch): bit-exactness vs the twin, the conv / network / golden
# suites under the new default, bench line
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
(timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "conv1") > $O/pytest_conv1.log 2>&1; echo "pytest conv1 rc=$?" | tee -a $O/summary.log
tail -15 $O/pytest_conv1.log
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --dump-profile $O/prof.json) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench.log') if q.startswith('{')][-1]; d=json.loads(l)
print('fps %.1f ms %.3f dom %.4f ms frac %.3f step frac %.3f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['step_roofline']['frac']))
print(json.dumps(d.get('keypoint_match'))[:800])
for e in json.load(open('$O/prof.json'))['entries']:
    if e['layer'] in ('conv1_1+conv1_2','conv2_1','conv2_2'): print('%-16s %-34s %.4f ms'%(e['layer'],e['kernel'],e['total_ms']))
PY
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --engine-opt conv1_wino=0) > $O/bench_direct.log 2> $O/bench_direct.err
python - <<PY
import json
l=[q for q in open('$O/bench_direct.log') if q.startswith('{')][-1]; d=json.loads(l)
print('conv1_wino=0: fps %.1f ms %.3f'%(d['value'],d['ms_per_step']))
PY
(timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_winograd.py tests/test_gpu_network.py tests/test_gpu_reference_goldens.py tests/test_gpu_properties.py tests/test_gpu_selection.py tests/test_face_hand.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -15 $O/pytest.log
