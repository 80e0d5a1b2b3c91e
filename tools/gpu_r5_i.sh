# Round-5 GPU call I: separable cubic resize (tests + precise timing), packed output transform A/B
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_precise.py -m gpu -x -q) > $O/pytest_precise.log 2>&1; echo "pytest precise rc=$?" | tee -a $O/summary.log
tail -4 $O/pytest_precise.log
(timeout 600 python tools/kernel_variants.py time --steps 5 --json $O/pkout.json) 2>&1 | tee $O/pkout.log
(timeout 600 python tools/kernel_variants.py time --steps 5 --json $O/pkout2.json) 2>&1 | tee $O/pkout2.log
cd /tmp; (timeout 400 rocprofv3 --kernel-trace --stats -d $O/rp_precise -o precise --output-format csv -- python $R/tools/precise_bench_driver.py) > $O/rp_precise.log 2>&1; cd $R
tail -2 $O/rp_precise.log
rm -f $O/rp_precise/*trace.csv $O/rp_precise/*/*trace.csv
grep -h "resize_cubic\|pp_limbs\|pp_group" $O/rp_precise/*stats.csv $O/rp_precise/*/*stats.csv 2>/dev/null | cut -c1-60,160-260
