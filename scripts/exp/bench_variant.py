#!/usr/bin/env python3
"""bench.py against an experimental build: scripts/exp/bench_variant.py <name> [bench.py flags]  (lib_exp/<name>/libmvhmr_unproject.so)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multiviewhmr_amd import _capi
name = sys.argv.pop(1)
if name != "shipped":
    _capi.LIB_PATH = os.path.join(ROOT, "multiviewhmr_amd", "lib_exp", name, "libmvhmr_unproject.so")
import bench
bench.main()
# experiment builds with phase timers (MVHMR_EXP_BWD bit 6): print them after the run
import ctypes
lib = _capi.lib()
if hasattr(lib, "mvhmr_exp_timers_read"):
    buf = (ctypes.c_ulonglong * 8)()
    lib.mvhmr_exp_timers_read(buf, 0)
    waves = max(1, buf[6]); tot = sum(buf[i] for i in range(6))
    names = ["resample issued", "scales + DMA request + adds + Jacobian issued", "barrier 1", "flush issued", "vmcnt wait", "barrier 2"]
    sys.stderr.write("phase timers (s_memtime ticks per wave, whole quad loop; %d waves)\n" % waves)
    for i in range(6):
        sys.stderr.write("  %-48s %10.0f  %5.1f %%\n" % (names[i], buf[i] / waves, 100.0 * buf[i] / tot))
if hasattr(lib, "mvhmr_exp_fwd_timers_read"):
    buf = (ctypes.c_ulonglong * 8)()
    lib.mvhmr_exp_fwd_timers_read(buf, 0)
    waves = max(1, buf[6]); tot = sum(buf[i] for i in range(5))
    names = ["projections + tap records + wave boxes", "barrier (block boxes)", "windows, addresses, chunk table, first DMA issued",
             "window 0 landed + barrier", "quad loop"]
    sys.stderr.write("forward phase timers (s_memtime ticks per wave and brick; %d wave-bricks)\n" % waves)
    for i in range(5):
        sys.stderr.write("  %-52s %10.0f  %5.1f %%\n" % (names[i], buf[i] / waves, 100.0 * buf[i] / tot))
