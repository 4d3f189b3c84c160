"""Lab: cProfile of steady training epochs (host side): epoch_cprofile.py <sup|semisup> [epochs]"""
import cProfile, io, json, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tools import bench_stages, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "semisup"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
x, y = synth.rmnist_like(600, 100, seed=1337, device=dev)
hp = json.load(open(os.path.join(ROOT, "tests", "golden", "hyperparameters.json")))["srmnist_manifold_semisupervised"]
bench_stages.training_stage(x, y, hp, dev, semisup=(mode == "semisup"), epochs=2)      # warm: compile / first captures
pr = cProfile.Profile()
pr.enable()
out = bench_stages.training_stage(x, y, hp, dev, semisup=(mode == "semisup"), epochs=epochs)
pr.disable()
print(json.dumps(dict(mode=mode, epoch_ms_all=out["epoch_ms_all"])))
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print("\n".join(l[:200] for l in s.getvalue().splitlines()))
