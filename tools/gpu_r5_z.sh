# Round-5 GPU call Z: the clustered transform on ALL geometries (PMX_WINO_VCL_GEOMS=15) -- possible once the unroller is told to honour the
# full unrolls of the phase bodies (-mllvm -pragma-unroll-threshold: the "3 KB of scratch" of the rectangle / multi-slab forms was a
# declined unroll) -- against the committed default; then the variant in the product's place: bit-exactness suites and the full bench line
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05z; mkdir -p $O; cd $R
(timeout 300 python tools/kernel_variants.py time --steps 5 --json $O/vclall.json) 2>&1 | tee $O/vclall.log
cp tools/_build/libpose_var_vcl15.so chainer_realtime_multi-person_pose_estimation_amd/csrc/libpose_mi355x.so
(timeout 600 python bench.py) > $O/bench.log 2> $O/bench.err; echo "bench (vcl15 library) rc=$?" | tee -a $O/summary.log
python - <<PY
import json
l=[q for q in open('$O/bench.log') if q.startswith('{')][-1]; d=json.loads(l)
print('fps %.1f ms %.3f dom %.4f frac %.3f step %.3f | single %.3f ms | precise %.2f ms batch8 %.2f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['step_roofline']['frac'],d['single_image']['ms_per_call'],d['precise']['ms_per_image'],d['precise']['batch8']['ms_per_image']))
r=d['rect_368x496']
print('rect', {k:(v['value'], v['rate_per_pixel_vs_square'], v['dominant_kernel']['issued_frac']) for k,v in r.items()})
print('match', d['keypoint_match']['frames_with_identical_peak_indices'], d['keypoint_match']['network_vs_order_defined_oracle'], d['precise']['keypoint_match_vs_precise_ref']['mismatching_peaks'])
PY
(timeout 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv.py tests/test_gpu_network.py tests/test_gpu_reference_goldens.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest (vcl15 library) rc=$?" | tee -a $O/summary.log
tail -4 $O/pytest.log
