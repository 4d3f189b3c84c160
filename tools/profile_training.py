#!/usr/bin/env python
"""Epochs of manifold_informed_train at C3 / C4 size for `rocprofv3 --kernel-trace --stats`: profile_training.py <sup|semisup> <epochs>.
Run twice with different epoch counts; tools/summarize_training_profile.py takes the difference of the two kernel-stats files as
the kernel time / launch count of (E2 - E1) epochs (setup -- k-NN, graph, first-call captures -- cancels)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tools import bench_stages, synth

mode, epochs = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
x, y = synth.rmnist_like(600, 100, seed=1337, device=dev)
hp = json.load(open(os.path.join(ROOT, "tests", "golden", "hyperparameters.json")))["srmnist_manifold_semisupervised"]
out = bench_stages.training_stage(x, y, hp, dev, semisup=(mode == "semisup"), epochs=epochs)
print(json.dumps(dict(mode=mode, epochs=epochs, epoch_ms_all=out["epoch_ms_all"], epoch_ms=out["epoch_ms"])))
