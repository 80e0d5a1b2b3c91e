R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_bench_launch.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
(timeout 300 python bench.py --steps 20 --warmup 3 --force-gather --no-cpu-baseline --no-extras) > $O/bench_fg.log 2> $O/bench_fg.err; echo "bench fg rc=$?" | tee -a $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench_fg.log') if q.startswith('{')][-1]; d=json.loads(l)
print('force-gather: fps %.1f ms %.3f gather_ms %.3f host_wait %.3f coll/step %.2f'%(d['value'],d['ms_per_step'],d['gather_ms_per_step_rank0'],d['host_wait_for_gpu_ms_per_step_rank0'],d['collectives_per_step']))
PY
tail -2 $O/pytest.log
