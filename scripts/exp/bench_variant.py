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
