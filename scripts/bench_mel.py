#!/usr/bin/env python3
"""Times calc_power_spectrogram on 32 resident 10 s clips (HIP events)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(bench.measure_mel(torch.device('cuda:0'), 32, 10.0, int(sys.argv[1]) if len(sys.argv) > 1 else 80, reps=20))
