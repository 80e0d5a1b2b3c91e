R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15) > $O/pytest_gpu.log
(timeout 600 python bench.py --steps 5 --warmup 2 --dump-profile $O/prof_bench.json) > $O/bench.log 2>&1
(timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --batch 8 --backend gloo --no-profile) > $O/bench_2rank_gloo.log 2>&1
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1
tail -3 $O/pytest_gpu.log; tail -1 $O/bench.log; tail -2 $O/bench_2rank_gloo.log; tail -1 $O/smoke.log
