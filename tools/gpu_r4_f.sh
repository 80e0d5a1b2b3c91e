# Round-4 GPU call F: the buffer-store epilogue of the Winograd kernels: bit-exactness tests, bench, block timing
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv.py tests/test_gpu_network.py tests/test_gpu_properties.py tests/test_gpu_reference_goldens.py tests/test_gpu_selection.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
for i in 1 2; do
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --dump-profile $O/prof.json) > $O/bench$i.log 2> $O/bench$i.err; echo "bench rc=$?" | tee -a $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench$i.log') if q.startswith('{')][-1]; d=json.loads(l)
print('fps %.1f ms %.3f dom %.4f ms frac %.3f step frac %.3f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['step_roofline']['frac']))
PY
done
(timeout 600 python tools/block_timing.py --ks 7 --cin 128 --batch 64 --json $O/bt_7x7.json) > $O/bt_7x7.log 2>&1
(timeout 600 python tools/block_timing.py --ks 3 --cin 64 --cout 128 --hw 184 --batch 32 --json $O/bt_conv2_1.json) > $O/bt_conv2_1.log 2>&1
(timeout 600 python tools/block_timing.py --ks 3 --cin 256 --cout 256 --hw 92 --batch 32 --json $O/bt_conv3_2.json) > $O/bt_conv3_2.log 2>&1
tail -3 $O/pytest.log
python - <<'PY'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04f'
for f in ('bt_7x7','bt_conv2_1','bt_conv3_2'):
    try:
        d=json.load(open(O+'/'+f+'.json'))
        print(f,'layer_ms',round(d['layer_ms'],4),{k[:8]:round(v[0],2) for k,v in d['phases_us_mean_p10_p90'].items()},'gap',round(d['handover_gap_us_mean_p10_p90'][0],2))
    except Exception as e: print(f,e)
for e in json.load(open(O+'/prof.json'))['entries']:
    if e['kernel'].startswith('conv_wino'): print('%-14s %-30s %.4f ms'%(e['layer'],e['kernel'],e['total_ms']))
PY
