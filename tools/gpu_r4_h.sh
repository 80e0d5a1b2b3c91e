R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R
for G in 0 1 2 3; do
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --engine-opt wino_tail_g=$G --dump-profile $O/prof_g$G.json) > $O/bench_g$G.log 2> $O/bench_g$G.err
python - <<PY
import json
l=[q for q in open('$O/bench_g$G.log') if q.startswith('{')][-1]; d=json.loads(l)
t=[e for e in json.load(open('$O/prof_g$G.json'))['entries'] if e['kernel'].endswith(':units')]
print('tail_g $G: fps %.1f ms %.3f | units total %.3f ms | Mconv1 %.4f Mconv2 %.4f conv4_2 %.4f'%(d['value'],d['ms_per_step'],sum(e['total_ms'] for e in t),[e['total_ms'] for e in t if e['layer']=='Mconv1_stage2'][0],[e['total_ms'] for e in t if e['layer']=='Mconv2_stage2'][0],[e['total_ms'] for e in t if e['layer']=='conv4_2'][0]), [e['kernel'] for e in t if e['layer'] in ('Mconv1_stage2','Mconv2_stage2','conv4_2')])
PY
done
