#!/usr/bin/env python3
"""the GPU tests against an experimental build: scripts/exp/pytest_variant.py <name> [pytest args]  (lib_exp/<name>/libmvhmr_unproject.so)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multiviewhmr_amd import _capi
name = sys.argv.pop(1)
_capi.LIB_PATH = os.path.join(ROOT, "multiviewhmr_amd", "lib_exp", name, "libmvhmr_unproject.so")
import pytest
sys.exit(pytest.main(sys.argv[1:]))
