# Round-5 GPU call K: detect_precise, chip-filling scales alone + small scales in flight together: tests, timing
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_precise.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -4 $O/pytest.log
(timeout 600 python tools/precise_bench_driver.py) > $O/precise.log 2>&1; tail -1 $O/precise.log | cut -c1-400
(timeout 600 python tools/precise_bench_driver.py) > $O/precise2.log 2>&1; tail -1 $O/precise2.log | cut -c1-200
