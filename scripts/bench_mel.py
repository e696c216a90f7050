#!/usr/bin/env python3
"""Times calc_power_spectrogram on resident 10 s clips (HIP events): bench_mel.py [n_mels] [batch]."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n_mels = int(sys.argv[1]) if len(sys.argv) > 1 else 80
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
print(bench.measure_mel(torch.device('cuda:0'), batch, 10.0, n_mels, reps=20))
