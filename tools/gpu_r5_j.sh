# Round-5 GPU call J: detect_precise with the scales in flight on four lanes: tests, timing (bench precise object), conv1_wino small maps
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05j; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_precise.py tests/test_gpu_conv.py -m gpu -x -q -k "precise or cubic or conv1") > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -6 $O/pytest.log
(timeout 600 python tools/precise_bench_driver.py) > $O/precise.log 2>&1; tail -1 $O/precise.log | cut -c1-600
(timeout 300 python - <<'PY'
import importlib, json, sys, time
sys.path.insert(0, '.')
import numpy as np, bench
PD = importlib.import_module(bench.PKG + '.pose_detector'); W_ = importlib.import_module(bench.PKG + '.weights')
H, W = 482, 642
img = np.random.default_rng(55).integers(0, 256, (H, W, 3), dtype=np.uint8)
det = PD.PoseDetector(weights=W_.synthetic_weights(0), device=0, precise=True, max_size=(736, 984))
for lanes in (4, 1, 2, 4, 1):
    det.engine.set_option('precise_lanes', lanes)
    for _ in range(2):
        try: det._detect_precise_device(img, fetch_maps=False)
        except IndexError: pass
    t0 = time.perf_counter()
    for _ in range(5):
        try: det._detect_precise_device(img, fetch_maps=False)
        except IndexError: pass
    print('lanes', lanes, 'ms per image %.2f' % ((time.perf_counter() - t0) / 5 * 1e3))
PY
) 2>&1 | tail -6
