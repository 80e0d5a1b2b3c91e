# Round-4 GPU call D: persistent form of the 3x3 Winograd launches (option wino_persist) -- conv tests under it, A/B timing; batched precise test + bench leg
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
(PMX_WINO_PERSIST=1 timeout 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv.py tests/test_gpu_network.py -m gpu -x -q) > $O/pytest_persist.log 2>&1; echo "pytest persist rc=$?" | tee -a $O/summary.log
(timeout 600 python -m pytest tests/test_precise.py -m gpu -x -q) > $O/pytest_precise.log 2>&1; echo "pytest precise rc=$?" | tee -a $O/summary.log
for X in 0 1 0 1; do
  (timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --engine-opt wino_persist=$X --dump-profile $O/prof_p$X.json) > $O/bench_p$X.log 2> $O/bench_p$X.err; echo "bench persist=$X rc=$?" | tee -a $O/summary.log
  python - <<PY
import json
l=[q for q in open('$O/bench_p$X.log') if q.startswith('{')][-1]; d=json.loads(l)
print('persist $X: %.1f fps %.3f ms'%(d['value'],d['ms_per_step']))
PY
done
python - <<'PY'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04d'
a=json.load(open(O+'/prof_p0.json'))['entries']; b=json.load(open(O+'/prof_p1.json'))['entries']
kb={e['layer']+'|'+e['kernel']:e for e in b}
for e in a:
    k=e['layer']+'|'+e['kernel']
    if k in kb and e['kernel'].startswith('conv_wino_f2x2_3x3') and ':' not in e['kernel']:
        print('%-14s %-28s plain %.3f ms  persistent %.3f ms  %+.1f %%'%(e['layer'],e['kernel'],e['total_ms'],kb[k]['total_ms'],(kb[k]['total_ms']/e['total_ms']-1)*100))
PY
(timeout 600 python - <<'PY'
import importlib, json, sys
sys.path.insert(0, '.')
import bench
out = bench.precise_mode(importlib.import_module(bench.PKG + '.weights'), 0, with_oracle=False)
print(json.dumps({k: out[k] for k in ('ms_per_image', 'batch8', 'peaks', 'people')}))
PY
) > $O/precise_batch.log 2>&1; tail -2 $O/precise_batch.log
tail -3 $O/pytest_persist.log; tail -3 $O/pytest_precise.log
