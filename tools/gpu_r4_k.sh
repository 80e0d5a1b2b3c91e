# merged tails: tests, then A/B of the whole step with the layer profile (tails = ":units" + ":combine" entries)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; mkdir -p $O; cd $R
(timeout 600 python -m pytest tests/test_gpu_winograd.py -x -q -k "merged or tail") > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for m in 1 0; do
  timeout 200 python tools/profile_driver.py --batch 32 --steps 5 --opt wino_tail_merge=$m --profile-json $O/prof_m$m.json | head -1
  python - <<PY
import json
d=json.load(open('$O/prof_m$m.json'))
t=sum(e['avg_ms'] for e in d['entries'] if e['kernel'].endswith(':units')); c=sum(e['avg_ms'] for e in d['entries'] if e['kernel'].endswith(':combine'))
print('merge=$m units %.3f ms combine %.3f ms total %.3f' % (t, c, sum(e['avg_ms'] for e in d['entries'])))
for e in d['entries']:
    if e['layer'] in ('Mconv1_stage2','Mconv2_stage2','conv4_2','conv5_1_CPM') and ':' in e['kernel']: print('  ', e['layer'], e['kernel'], '%.4f' % e['avg_ms'])
PY
done
(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.log
