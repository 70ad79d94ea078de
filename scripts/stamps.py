"""Debug helper: per-segment wave time of the forward brick kernel (stamp build, MVHMR_ABL=20)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVHMR_FWD2"] = sys.argv[1] if len(sys.argv) > 1 else "1,1,1180160"
import torch
import bench
from multiviewhmr_amd import _capi
L = _capi.lib()
L.mvhmr_debug_stamps3.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-backward", "--no-check", "--steps", "3", "--warmup", "1"]
L.mvhmr_debug_stamps3(None, 1)
bench.main()
L.mvhmr_debug_stamps3(buf, 0)
v = np.array(list(buf), dtype=np.float64)
waves = v[15]
names = ["top: vmcnt wait", "barrier", "dma issue", "job0 reads+aggregate", "job0 store", "job0 folds", "job1 reads+agg+store", "job1 folds"]
tot = v[:8].sum()
print("waves %d, total stamped cycles per wave %.0f" % (waves, tot / waves))
for n, x in zip(names, v[:8]):
    print("  %-18s %8.0f cycles per wave (%.1f per quad)  %5.1f %%" % (n, x / waves, x / waves / 63, 100 * x / tot))
