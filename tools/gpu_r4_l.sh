# phase barrier moved to step 30 (next phase's first fragments requested behind it): Winograd tests, then the step with the layer profile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4l; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_winograd.py -x -q) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 200 python tools/profile_driver.py --batch 32 --steps 5 --profile-json $O/prof.json | head -1
python - <<PY
import json, sys
sys.path.insert(0, 'tools')
from kernel_variants import group
d=json.load(open('$O/prof.json')); g={}
for e in d['entries']:
    k=group(e['layer'], e['kernel']); g[k]=g.get(k,0)+e['avg_ms']
print(' '.join('%s %.3f' % kv for kv in sorted(g.items())), 'sum %.3f' % sum(g.values()))
for e in d['entries']:
    if e['layer'] in ('conv2_1','conv2_2','conv3_1','conv3_2','conv4_2','Mconv1_stage2','Mconv2_stage2','conv5_1_CPM'): print('  ', e['layer'], e['kernel'], '%.4f' % e['avg_ms'])
PY
(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.log
