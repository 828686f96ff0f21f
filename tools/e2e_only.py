#!/usr/bin/env python3
"""bench.py's end-to-end section alone (one batch at a time, two host threads, anx_pipeline): e2e_only.py"""
import argparse
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tools.fresh_batch import setup  # noqa: E402

torch, A, model, queries, params, paths = setup()
args = argparse.Namespace(queries=len(queries))
e = bench.e2e_of(args, model, queries, params, torch.cuda.current_stream().cuda_stream, torch)
e.pop("pipelined_last", None)
print(json.dumps({k: v for k, v in e.items() if k != "what"}))
