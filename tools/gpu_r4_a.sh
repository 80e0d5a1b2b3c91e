# Round-4 GPU call A: the new tests (census, pipelined gather, bench launcher), the 512-frame margin census, the bench line.
# Usage: gpurun --timeout 1500 -- 'bash tools/gpu_r4_a.sh'
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.log
(timeout 900 python -m pytest tests/test_gpu_census.py tests/test_gpu_dist.py tests/test_bench_launch.py tests/test_precise.py "tests/test_gpu_winograd.py::test_batch_32_default_path_vs_cpu_oracle_and_single_images" -m gpu -x -q -s) > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/summary.log
(timeout 900 python tools/parity_census.py --frames 512 --bf16x3 --out $O/parity_census.json) > $O/census.log 2> $O/census.err; echo "census rc=$?" | tee -a $O/summary.log
(timeout 600 python bench.py --steps 10 --warmup 3 --dump-profile $O/prof_bench.json) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.log
tail -5 $O/pytest_new.log; tail -c 1500 $O/census.log; tail -c 400 $O/bench.err
