"""Lab: the 100-column CG solve of bench.py's `cg_multi_rhs` block alone (for rocprofv3 --kernel-trace --stats).
Usage (GPU box):  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mrhs -- python3 tools/lab_multi_rhs.py [columns]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    cols = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    import argparse
    args = argparse.Namespace(workload="c3", nodes=0, s5_order="morton")
    wl = bench.build_workload(args, torch.device("cuda:0"), 0, 1)
    out = bench.multi_rhs_solve(wl, columns=cols)
    print(out)


if __name__ == "__main__":
    main()
