# last refresh of round 4 (new Winograd weight layout): GPU suite, bench line, rocprofv3 kernel stats of the bench command, force-gather line
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
(timeout 600 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -2 $O/pytest_gpu.log
(timeout 300 python bench.py --steps 20 --warmup 3 --dump-profile $O/prof_bench.json) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; cut -c1-230 $O/bench.log
(timeout 200 python bench.py --steps 5 --warmup 2 --force-gather --no-cpu-baseline --no-extras) > $O/bench_force_gather.log 2> $O/bench_force_gather.err; echo "force-gather rc=$?"
cd /tmp; rm -rf $O/rp_bench
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/rp_bench -o bench --output-format csv -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras) > $O/rp_bench.log 2>&1; echo "rocprof rc=$?"
rm -f $O/rp_bench/*trace.csv
