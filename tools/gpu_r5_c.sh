# Round-5 GPU call C: transposed tiles with early bias (3x3: through LDS, 7x7: loaded in pass 2b) vs the round-4 library; planar batched cubic
# resizes + cached tables in detect_precise: tests, per-layer A/B, bench with the precise object
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_precise.py tests/test_gpu_winograd.py tests/test_gpu_conv.py tests/test_gpu_network.py tests/test_gpu_reference_goldens.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -5 $O/pytest.log
(timeout 600 python tools/kernel_variants.py time-conv --iters 10 --json $O/conv_ab.json) 2>&1 | tee $O/conv_ab.log
(timeout 600 python tools/kernel_variants.py time --steps 5 --json $O/variants.json) 2>&1 | tee $O/variants.log
(timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dump-profile $O/prof.json) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench.log') if q.startswith('{')][-1]; d=json.loads(l)
print('fps %.1f ms %.3f dom %.4f frac %.3f step %.3f | single %.3f ms | precise %.2f ms batch8 %.2f kernel_ms %.2f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['step_roofline']['frac'],d['single_image']['ms_per_call'],d['precise']['ms_per_image'],d['precise']['batch8']['ms_per_image'],d['precise'].get('kernel_ms_per_image',-1)))
print(json.dumps(d['precise'].get('keypoint_match_vs_precise_ref'))[:600])
PY
cd /tmp; (timeout 300 rocprofv3 --kernel-trace --stats -d $O/rp_precise -o precise --output-format csv -- python $R/tools/precise_bench_driver.py) > $O/rp_precise.log 2>&1; cd $R
rm -f $O/rp_precise/*/*trace.csv; head -12 $O/rp_precise/*/precise_kernel_stats.csv | cut -c1-150
