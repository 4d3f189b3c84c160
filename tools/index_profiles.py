#!/usr/bin/env python
"""Rewrite the round's section of profiles/INDEX.md from the committed summaries themselves, so that every number the index
quotes is the file's: tools/index_profiles.py r05 (after tools/assemble_profiles.py r05)."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"


def stats(name):
    return list(csv.DictReader(open(os.path.join(P, "%s_%s.csv" % (tag, name)))))


def row(rows, sub):
    r = next(r for r in rows if sub in r["Name"])
    return int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])


def J(name):
    return json.load(open(os.path.join(P, "%s_%s.json" % (tag, name))))


tr, s5 = J("pmc_traffic"), J("s5_pmc_traffic")
bt, s5b = J("bench_trace"), J("s5_bench_trace")
cxb = J("s5_complex_shift_bench")
sup, semi = J("training_supervised"), J("training_semisupervised")
B = bt["roofline"]["bytes_per_launch"]
B5 = s5b["roofline"]["bytes_per_launch"] if "roofline" in s5b else s5b["roofline_hbm"]["bytes_per_launch"]
ns, ns5 = tr["spmv_kernel_trace_mean_ns"], s5["spmv_kernel_trace_mean_ns"]
eig = stats("eigensolve_kernel_stats")
knn = stats("knn_kernel_stats")
cx = stats("s5_complex_shift_kernel_stats")
mt = row(eig, "spmm_mt_kernel")
gram = row(eig, "gram_mfma_kernel")
dm, sel, dt = row(knn, "dist_mfma_kernel<true>"), row(knn, "select_kernel"), row(knn, "dist_tile_kernel")
dmf, rg, bw = row(knn, "dist_mfma_kernel<false>"), row(knn, "regroup_kernel"), row(knn, "bound_wave_kernel")
cq, cu, cf = row(cx, "spmm_tile_q_kernel<1"), row(cx, "cx_update_kernel"), row(cx, "spmv_f64_kernel")
solve = open(os.path.join(P, tag + "_trace_solve.txt")).read().strip().splitlines()[-1]
kt = open(os.path.join(P, tag + "_pmc_knn_mfma.txt")).read()
busy = float(re.search(r"filtered_pass<true> SQ_VALU_MFMA_BUSY_CYCLES .* mean ([0-9.]+)", kt).group(1))
cuc = float(re.search(r"filtered_pass<true> SQ_BUSY_CU_CYCLES .* mean ([0-9.]+)", kt).group(1))
ks = open(os.path.join(P, tag + "_pmc_knn_select.txt")).read()
fetch = float(re.search(r"FETCH_SIZE launches \d+ median KB ([0-9.]+)", ks).group(1))
sel_us = float(re.search(r"durations: launches \d+ median us ([0-9.]+)", ks).group(1))
wr = float(re.search(r"WRITE_SIZE launches \d+ median KB ([0-9.]+)", ks).group(1))
mf = open(os.path.join(P, tag + "_pmc_mfma.txt")).read()
kb = re.search(r"knob 0: ([0-9.]+) us per launch = ([0-9.]+) TFLOP/s", mf)
kbusy = re.search(r"matrix pipe busy (\d+) % of the launch", mf)
wide = open(os.path.join(P, tag + "_pmc_spmm_wide.txt")).read()
miss = re.search(r"TCC_MISS_sum\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)", wide)
sec = """Round 5 (`%(tag)s_*`), one MI355X.  Every file: `tools/retake_profiles.sh %(tag)s` on the GPU box, then `python tools/assemble_profiles.py %(tag)s` and
`python tools/index_profiles.py %(tag)s` here (this section is generated from the files: every number below is read from them).  Source
hash of the tree they were taken on: `%(hash)s` (`bench.py` quotes a profile in `roofline.frac` only while it matches and the live
figure lies within -15 / +25 %% of it -- `profile_age_ok`).  New this round: the k-NN search (kernel mix + both counter passes; since the candidate filter: the filtered key pass, the sample keys, regroup and bounds), the S5 solve as the
library runs it by default (complex-shift COCG), the two training epochs, the wide SpMM on the chain-relabelled matrix.

| file | recipe | backs |
|---|---|---|
| `%(tag)s_kernel_stats.csv`, `%(tag)s_bench_trace.json`, `%(tag)s_pmc_traffic.json` | `tools/profile.sh %(tag)s` = `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras`, then the same command under `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes), `tools/summarize_profile.py` | `roofline.in_graph_profile` of the bench line: `spmv_tile_kernel<false,256>` %(ns).1f ns mean over %(nl)d launches -> %(mb).2f MB algorithmic / %(us).4f us = %(tbs).2f TB/s = %(frac).4f of 8 TB/s on the box of this pass (boxes of the pool: 5.4-6.2 us); `roofline.traffic` %(traffic)d bytes per launch (2 x FETCH_SIZE + WRITE_SIZE); the kernel mix of a solve |
| `%(tag)s_trace_solve.txt` | `tools/trace_solve.py` over the same kernel trace (the last graph-replayed solve that came out whole under the profiler) | %(solve)s |
| `%(tag)s_s5_kernel_stats.csv`, `%(tag)s_s5_bench_trace.json`, `%(tag)s_s5_pmc_traffic.json` | `tools/profile.sh %(tag)s_s5 --workload s5 --classic-cg` (the C = 1 streaming kernel is what `roofline_hbm` prices, so this pass keeps the classic CG that runs it twice per iteration) | `roofline_hbm`: %(ns5).1f ns mean over %(nl5)d launches -> %(frac5).3f of 8 TB/s in-graph under the profiler (%(mb5).0f MB algorithmic); counter traffic %(t5).1f MB per launch (the x dictionary hits in L2) |
| `%(tag)s_s5_complex_shift_kernel_stats.csv`, `%(tag)s_s5_complex_shift_bench.json` | `rocprofv3 --kernel-trace --stats -- python3 bench.py --workload s5 --steps 20 --warmup 3 --no-cpu-baseline --no-extras` (the default solver for I + s Q^2 systems: COCG on the complex factor I + i sigma B) | the S5 solve at %(cxms).1f ms under the profiler (round 4: 133 ms): `spmm_tile_q_kernel<1,false>` (4 interleaved columns: re, im of two products) %(cq1).1f us x %(cq0)d, `cx_update_kernel` %(cu1).1f us x %(cu0)d, the fp64 residual checks `spmv_f64_kernel` %(cf1).0f us x %(cf0)d; the bench line of that run |
| `%(tag)s_eigensolve_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats -- python3 tools/time_eigen.py 100 1e-6` | three 60k eigensolves on the chain-relabelled matrix: `spmm_mt_kernel<false>` %(mt1).1f us x %(mt0)d under the profiler (r04: 70.3 us x 819; unprofiled 34.5 us), %(mt2).1f %% of a run that also builds the graph; `gram_mfma_kernel` %(g1).0f us x %(g0)d |
| `%(tag)s_pmc_spmm_wide.txt` | `tools/pmc_kernel.sh` over `tools/lab/spmm_one.py` (C = 128, 60k graph): the matrix-core tile kernel on the chain-relabelled matrix, the same kernel on the matrix in the given order (`MGP_NO_CHAIN=1`), the gather kernel | launch cycles, L1 / L2 requests, instruction counts per launch; `TCC_MISS_sum` %(m1).0f / %(m2).0f / %(m3).0f lines of 128 bytes (chain / given / gather; compulsory for the chain-relabelled product: image + X + Y = 96.4 MB = 753 k lines -> %(mr).2f x); docs/kernels/spmm.md round 5 reads them (the tile kernel is bound by matrix-pipe work per distinct column and a per-wave fixed cost -- halving the distinct columns per tile nearly halves the launch) |
| `%(tag)s_knn_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats -- python3 tools/time_knn.py` (60k x 784: three searches with the candidate filter, the graph build, three with every tile computed, three with direct-difference keys on the slab) | `stages.knn`: `dist_mfma_kernel<true>` (the filtered key pass; upper-triangle and every-tile launches mixed) %(dm1).0f us x %(dm0)d, `dist_mfma_kernel<false>` (keys to the sampled points) %(dmf1).0f us x %(dmf0)d, `select_kernel` %(sel1).0f us x %(sel0)d, `regroup_kernel` %(rg1).0f us x %(rg0)d, `bound_wave_kernel` %(bw1).0f us x %(bw0)d, `dist_tile_kernel<true>` (the direct-difference leg) %(dt1).0f us x %(dt0)d |
| `%(tag)s_pmc_knn_mfma.txt` | `tools/pmc_knn.sh` | `dist_mfma_kernel`: SQ_VALU_MFMA_BUSY_CYCLES %(busy).1f M / 1024 SIMDs of SQ_BUSY_CU_CYCLES %(cuc).1f M / 256 CUs -> matrix pipe busy %(bp).1f %% of the launch |
| `%(tag)s_pmc_knn_select.txt` | `tools/pmc_select.sh` | `select_kernel` (one launch per 60k-row search, candidate lists): FETCH_SIZE median %(fe).3f GB (x 2 on gfx950 = %(fe2).2f GB) in %(su).0f us = %(sbw).1f TB/s of list + candidate-row reads from memory; WRITE_SIZE %(wr).1f MB |
| `%(tag)s_pmc_mfma.txt` | `tools/lab/pmc_kbres.sh`, `tools/lab/trace_kbres.sh`, `tools/pmc_mfma.sh` | `mgp_kernel_block` at 600 x 60000 x 100 (unchanged kernel, retaken): matrix pipe busy %(kbusy)s %% of the launch, %(kb1)s us under the profiler = %(kb2)s TFLOP/s |
| `%(tag)s_training_supervised.json`, `%(tag)s_training_semisupervised.json` | `tools/profile_training.sh %(tag)s` = `rocprofv3 --kernel-trace --stats -- python3 tools/profile_training.py <mode> <epochs>` at two epoch counts, differenced (`tools/summarize_training_profile.py`) | `stages.train_epoch_*.kernel_time_profile` / `.kernel_share`: kernel time per epoch %(sk).2f ms in %(sl).0f launches (supervised, 60k x 784, all labelled) and %(mk).1f ms in %(ml).0f launches (semi-supervised, 6k labelled), with the top kernels of each |

""" % dict(tag=tag, hash=tr["source_hash"], ns=ns, nl=tr["spmv_kernel_trace_launches"], mb=B / 1e6, us=ns / 1e3, tbs=B / ns / 1e3,
           frac=B / ns / 8e3, traffic=tr["spmv_hbm_bytes_per_launch"], solve=solve, ns5=ns5, nl5=s5["spmv_kernel_trace_launches"],
           frac5=B5 / ns5 / 8e3, mb5=B5 / 1e6, t5=s5["spmv_hbm_bytes_per_launch"] / 1e6, cxms=cxb["ms_per_step"], cq0=cq[0], cq1=cq[1],
           cu0=cu[0], cu1=cu[1], cf0=cf[0], cf1=cf[1], mt0=mt[0], mt1=mt[1], mt2=mt[2], g0=gram[0], g1=gram[1],
           m1=float(miss.group(1)), m2=float(miss.group(2)), m3=float(miss.group(3)), mr=float(miss.group(1)) / 753125.0,
           dm0=dm[0], dm1=dm[1], dmf0=dmf[0], dmf1=dmf[1], rg0=rg[0], rg1=rg[1], bw0=bw[0], bw1=bw[1], sel0=sel[0], sel1=sel[1], dt0=dt[0], dt1=dt[1], busy=busy / 1e6, cuc=cuc / 1e6,
           bp=100.0 * (busy / 1024) / (cuc / 256), fe=fetch * 1024 / 1e9, fe2=2 * fetch * 1024 / 1e9, su=sel_us,
           sbw=2 * fetch * 1024 / sel_us / 1e6, wr=wr * 1024 / 1e6, kbusy=kbusy.group(1) if kbusy else "?", kb1=kb.group(1) if kb else "?",
           kb2=kb.group(2) if kb else "?", sk=sup["kernel_ms_per_epoch"], sl=sup["launches_per_epoch"], mk=semi["kernel_ms_per_epoch"],
           ml=semi["launches_per_epoch"])
idx = os.path.join(P, "INDEX.md")
s = open(idx).read()
i = s.index("Round 5 (`%s_*`)" % tag)
j = s.index("Round 4 (`r04_*`)")
open(idx, "w").write(s[:i] + sec + s[j:])
print(sec[:1200])
