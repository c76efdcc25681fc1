"""gpurun_out/seeds (tools/seed_spread.sh) -> profiles/<tag>_procedural_end_to_end.md: mean and spread of PSNR / size over
seeds and schedules, and what that spread means for a +-0.05 dB / +-0.5 % comparison.   python tools/seed_spread_summary.py r05"""
import math
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join(ROOT, "gpurun_out", "seeds")
rows = {}
for thread in (1, 0):
    path = os.path.join(src, f"results_t{thread}.txt")
    if not os.path.exists(path):
        continue
    for seed, line in zip((42, 43, 44), [l for l in open(path) if l.strip()]):
        c = line.rstrip("\n").split("\t")
        # columns (train_CNC_nerf_synthetic.py:562-613): scene psnr lpips -ssim psnr_c lpips_c -ssim_c est_MB coded_MB
        # mlp_MB ctx_MB occ_MB | 13 mlp13_MB psnr13 lpips13 ssim13 total_MB | train_s enc_s dec_s
        rows[(thread, seed)] = dict(psnr=float(c[1]), psnr_dec=float(c[4]), est_MB=float(c[7]), coded_MB=float(c[8]),
                                    psnr13=float(c[14]), total_KB=float(c[17]) * 1024.0, train_s=float(c[18]),
                                    enc_s=float(c[19]), dec_s=float(c[20]), line=line.rstrip("\n"))


def stats(vals):
    n = len(vals)
    m = sum(vals) / n
    sd = math.sqrt(sum((v - m) ** 2 for v in vals) / (n - 1)) if n > 1 else float("nan")
    return m, sd, min(vals), max(vals)


out = [f"# {TAG} — the whole protocol at full size on the procedural scene: 3 seeds x 2 schedules (one MI355X)", "",
       "`python -m cnc_amd.train --dataset procedural --image_size 400 --n_features 8 --sample_num 150000 --max_steps 5000",
       "--test_views 8 --seed S` (`tools/seed_spread.sh`; configs[2] composition: 12x3-D levels T=2^19 + 3x4 planes T=2^17, F=8,",
       "lmbda=2e-3; the nerf_synthetic images are not available offline, so the scene is the built-in analytic ball: these numbers",
       "say the pipeline works end to end at full size on the round's FINAL code and how far two runs of ONE configuration are",
       "apart — they are NOT the chair figures of BASELINE.md).  `threaded` = the default schedule (context pass on a second host",
       "thread and side stream), `sequential` = `CNC_CTX_THREAD=0` (the reference's order of random draws).", "",
       "| schedule | seed | PSNR | PSNR decoded | PSNR 13-bit MLP | estimate / coded MB | total KB | train s (ms/step) | enc / dec s |",
       "|---|---|---|---|---|---|---|---|---|"]
for (thread, seed), r in sorted(rows.items(), key=lambda kv: (-kv[0][0], kv[0][1])):
    out.append(f"| {'threaded' if thread else 'sequential'} | {seed} | {r['psnr']:.2f} | {r['psnr_dec']:.2f} | {r['psnr13']:.2f} | "
               f"{r['est_MB']:.4f} / {r['coded_MB']:.4f} | {r['total_KB']:.1f} | {r['train_s']:.1f} ({r['train_s'] / 5.0:.2f}) | "
               f"{r['enc_s']:.2f} / {r['dec_s']:.2f} |")
out.append("")
groups = {"threaded": [r for (t, _), r in rows.items() if t == 1], "sequential": [r for (t, _), r in rows.items() if t == 0],
          "all six": list(rows.values())}
out += ["| runs | PSNR 13-bit: mean +- sd (min .. max) | total KB: mean +- sd (min .. max) | sd of size / mean |", "|---|---|---|---|"]
sd_p = sd_k = mean_k = float("nan")
for name, g in groups.items():
    if len(g) < 2:
        continue
    mp, sp, lo_p, hi_p = stats([r["psnr13"] for r in g])
    mk, sk, lo_k, hi_k = stats([r["total_KB"] for r in g])
    out.append(f"| {name} ({len(g)}) | {mp:.2f} +- {sp:.2f} ({lo_p:.2f} .. {hi_p:.2f}) | {mk:.1f} +- {sk:.1f} ({lo_k:.1f} .. {hi_k:.1f}) | {100 * sk / mk:.2f} % |")
    if name == "all six":
        sd_p, sd_k, mean_k = sp, sk, mk
out.append("")
if rows and sd_p == sd_p:
    # two-sample comparison of means of n runs each: the difference has sd = s * sqrt(2 / n); resolve d at ~2 sd
    n_p = math.ceil(2 * (2 * sd_p / 0.05) ** 2)
    n_k = math.ceil(2 * (2 * (sd_k / mean_k) / 0.005) ** 2)
    out += ["**What a +-0.05 dB / +-0.5 % claim needs.**  The decoded tables are bit-identical to the trained ones (PSNR decoded = PSNR), so",
            "the run-to-run spread is all training: binarised tables under Adam are a chaotic system (DESIGN §6) and the seed, the",
            "schedule and the order of float atomics each give a different trajectory.  Comparing the means of n runs per side, the",
            f"difference of two such means has standard deviation s * sqrt(2 / n); to resolve 0.05 dB at two standard deviations with",
            f"s = {sd_p:.2f} dB takes **n = {n_p} runs per side**, and 0.5 % of the size with s = {100 * sd_k / mean_k:.2f} % takes **n = {n_k}**.",
            "A single chair run against a single reference run cannot decide the north_star's +-0.05 dB / +-0.5 % on this 5000-step",
            "schedule; the full 20000-step schedule anneals further (five learning-rate milestones) and should be re-measured the same",
            "way the day the dataset is available.", ""]
out += ["Results lines as the reference driver writes them (train_CNC_nerf_synthetic.py:562-613; LPIPS needs pretrained weights that are",
        "not available offline: NaN):", "```"]
out += [r["line"] for _, r in sorted(rows.items(), key=lambda kv: (-kv[0][0], kv[0][1]))] + ["```", ""]
open(os.path.join(ROOT, "profiles", f"{TAG}_procedural_end_to_end.md"), "w").write("\n".join(out))
print("\n".join(out))
