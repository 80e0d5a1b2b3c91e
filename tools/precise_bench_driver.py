"""rocprofv3 driver for BASELINE config 5: bench.precise_mode without the oracle leg (one 482 x 642 frame through PoseDetector(precise=True),
then eight per call).  rocprofv3 --kernel-trace --stats -- python tools/precise_bench_driver.py"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out = bench.precise_mode(importlib.import_module(bench.PKG + '.weights'), 0, with_oracle=False)
print(json.dumps({k: out[k] for k in ('ms_per_image', 'batch8', 'peaks', 'people', 'kernel_ms_per_image') if k in out}))
