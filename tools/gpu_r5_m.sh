# Round-5 GPU call M: sliced candidate scan of pp_limbs (goldens / fuzz / crowd / overflow under 0, 3, 8 slices), fused part summation of detect_precise
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R
(timeout 1200 python -m pytest tests/test_gpu_postprocess.py tests/test_gpu_properties.py tests/test_precise.py tests/test_gpu_reference_goldens.py tests/test_face_hand.py tests/test_gpu_census.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -6 $O/pytest.log
(timeout 600 python tools/precise_lanes_ab.py --json $O/lanes_ab.json) 2>&1 | tail -9
cd /tmp; (timeout 400 rocprofv3 --kernel-trace --stats -d $O/rp_precise -o precise --output-format csv -- python $R/tools/precise_bench_driver.py) > $O/rp_precise.log 2>&1; cd $R
rm -f $O/rp_precise/*trace.csv
grep -h "pp_limbs\|pp_group\|sum_parts\|resize_cubic" $O/rp_precise/*stats.csv | cut -c1-50,140-240
