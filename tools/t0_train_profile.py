#!/usr/bin/env python3
"""The T0_3B Conceptual-Captions training leg of bench.py alone (for rocprofv3 --kernel-trace): ViT-L/14 -> MLP mapper -> frozen T0_3B, B 64."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
a = argparse.Namespace(no_overlap=False, pipelined_update=False, no_roofline=True)
print(bench.t0_cc_train_leg("bf16", 10, 4, "cuda:0", a))
