# Round-5 soak: fresh-example fuzz of the post-process (sliced candidate scan) and of the conv kernels, run-to-run determinism of the batch path
# (conv1_wino_kernel in it), of single images and of detect_precise with four scales in flight
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05soak; mkdir -p $O; cd $R
(PMX_FUZZ=400 timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k hypothesis) > $O/fuzz_pp.log 2>&1; echo "fuzz post-process rc=$?" | tee -a $O/summary.log; tail -2 $O/fuzz_pp.log
(PMX_FUZZ=200 timeout 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv.py -m gpu -x -q -k "random or hypothesis or fuzz") > $O/fuzz_conv.log 2>&1; echo "fuzz conv rc=$?" | tee -a $O/summary.log; tail -2 $O/fuzz_conv.log
(timeout 600 python tools/soak.py --steps 200 --batch 32) > $O/soak32.log 2>&1; echo "soak 32 rc=$?" | tee -a $O/summary.log; tail -2 $O/soak32.log
(timeout 600 python tools/soak.py --steps 300 --batch 1) > $O/soak1.log 2>&1; echo "soak 1 rc=$?" | tee -a $O/summary.log; tail -2 $O/soak1.log
(timeout 600 python - <<'PY'
import importlib, sys, numpy as np
sys.path.insert(0, '.')
import bench
PD = importlib.import_module(bench.PKG + '.pose_detector'); W_ = importlib.import_module(bench.PKG + '.weights')
img = np.random.default_rng(55).integers(0, 256, (482, 642, 3), dtype=np.uint8)
det = PD.PoseDetector(weights=W_.synthetic_weights(0), device=0, precise=True, max_size=(736, 984))
ref = None; bad = 0
for i in range(40):
    det.engine.precise_begin(482, 642, 1)
    for s in (0.5, 1.0, 1.5, 2.0):
        m = s * 368 / 482
        det.engine.precise_add_scale(img, int(np.ceil(482 * m)), int(np.ceil(642 * m)))
    det.engine.precise_finish()
    paf, heat = det.engine.get_maps()
    if ref is None: ref = (paf.copy(), heat.copy())
    elif not (np.array_equal(paf, ref[0]) and np.array_equal(heat, ref[1])): bad += 1
print('precise soak: 40 sequences of four scales in flight, %d differing from the first' % bad)
PY
) > $O/soak_precise.log 2>&1; echo "soak precise rc=$?" | tee -a $O/summary.log; tail -1 $O/soak_precise.log
