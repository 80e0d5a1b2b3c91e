"""bench.precise_mode on its own (the `precise` object of the bench line), N times in one process."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
W_ = importlib.import_module(bench.PKG + '.weights')
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    o = bench.precise_mode(W_, 0, with_oracle=False)
    print(i, 'ms_per_image', round(o['ms_per_image'], 2), 'batch8', round(o['batch8']['ms_per_image'], 2), 'kernel_ms', round(o.get('kernel_ms_per_image', 0), 1),
          'per scale wall', [round(p['wall_ms'], 2) for p in o['per_scale_running_alone']]); sys.stdout.flush()
