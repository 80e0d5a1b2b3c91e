# Round-5 GPU call A: transposed accumulator tiles (16-byte channel-run stores) -- conv / network bit-exactness, A/B against the round-4 library, bench line
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv.py tests/test_gpu_network.py tests/test_gpu_reference_goldens.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -3 $O/pytest.log
(timeout 600 python tools/kernel_variants.py time --steps 5 --json $O/variants.json) 2>&1 | tee $O/variants.log
(timeout 600 python tools/kernel_variants.py time --steps 5 --json $O/variants2.json) 2>&1 | tee $O/variants2.log
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --dump-profile $O/prof.json) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench.log') if q.startswith('{')][-1]; d=json.loads(l)
print('fps %.1f ms %.3f dom %.4f ms frac %.3f step frac %.3f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['step_roofline']['frac']))
for e in json.load(open('$O/prof.json'))['entries']:
    if e['layer'] in ('conv1_1+conv1_2','conv2_1','conv2_2','conv3_1','conv3_2','conv3_4','conv4_2','conv5_1_CPM','Mconv1_stage2','Mconv2_stage2'): print('%-16s %-34s %.4f ms'%(e['layer'],e['kernel'],e['total_ms']))
PY
