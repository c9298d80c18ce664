#!/usr/bin/env python3
"""bench.py against a side build of the library (ND_LIB=path/to/lib.so), for same-box A/B comparisons:
    ND_LIB=tools/_build/libw2_0.so python tools/bench_ab.py --no-cpu --steps 20"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: F401
from noisediff_amd import _lib as L
if os.environ.get("ND_LIB"):
    L.load(os.environ["ND_LIB"])
import bench
bench.main()
