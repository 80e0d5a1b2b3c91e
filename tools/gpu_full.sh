# whole GPU test suite + bench.py (default flags) in one call.  usage: gpurun -- 'bash tools/gpu_full.sh <tag>'
TAG=${1:-full}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -40) > $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
(timeout 600 python bench.py --steps 5 --warmup 2 --dump-profile $O/prof_bench.json) > $O/bench.log 2> $O/bench.err; tail -c 3000 $O/bench.log; tail -5 $O/bench.err
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')") > $O/smoke.log 2>&1; tail -1 $O/smoke.log
