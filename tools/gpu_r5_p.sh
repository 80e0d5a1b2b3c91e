# Round-5 GPU call P: landscape / portrait census on the final kernels; bench.py --gpus 8 over gloo on the one GPU (the N > 1 path end to end at batch 32)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05p; mkdir -p $O; cd $R
(timeout 600 python tools/parity_census.py --frames 128 --h 368 --w 496 --out $O/census_368x496.json) > $O/c1.log 2>&1; echo "census 368x496 rc=$?" | tee -a $O/summary.log; tail -1 $O/c1.log
(timeout 600 python tools/parity_census.py --frames 96 --h 496 --w 368 --out $O/census_496x368.json) > $O/c2.log 2>&1; echo "census 496x368 rc=$?" | tee -a $O/summary.log; tail -1 $O/c2.log
(timeout 900 python bench.py --gpus 8 --backend gloo --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-extras) > $O/bench_8ranks_gloo.log 2> $O/bench_8ranks_gloo.err; echo "bench 8 ranks rc=$?" | tee -a $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench_8ranks_gloo.log') if q.startswith('{')][-1]; d=json.loads(l)
print('n_gpus',d['n_gpus'],'ranks_seen',d['ranks_seen'],'collectives/step',d['collectives_per_step'],'records',d['config']['records_gathered'],'value',round(d['value'],1),'gather ms',round(d['gather_ms_per_step_rank0'],3))
PY
