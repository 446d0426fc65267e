#!/usr/bin/env python3
"""Experiment (GPU box): the first-epoch leg of bench.py alone, for traces.
    python tools/r6/fresh_probe.py [nkeys] [minibatches] [percent]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

nkeys = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 12
pct = int(sys.argv[3]) if len(sys.argv) > 3 else 30
args = argparse.Namespace(seed=20260926, rows=50000, nnz_per_row=200, load_factor=0.5)
r = bench.fresh_table_leg(args, nkeys, nb, pct)
r.pop("what")
print(json.dumps(r))
