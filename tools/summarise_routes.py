"""gpurun_out/<tag>_routes/ (tools/collect_routes.sh) -> profiles/<tag>_headline_backward_routes.md: per scatter route of
the headline backward call its time, its kernels' average durations and their atomic requests (TCC_ATOMIC_sum), and the
distinct-cell counts per block size.      python tools/summarise_routes.py <tag>"""
import csv, glob, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out", f"{TAG}_routes")
routes = ["product", "runs", "merge", "cells", "carry"]
what = {"product": "the bench's call: merge kernel on the 10 coarse levels next to the binned 6 finest (side streams)",
        "runs": "k_grid_encode_bwd, 16 levels: runs of one cell along a ray, 256 samples per block",
        "merge": "k_grid_encode_bwd_merge, 16 levels: 1,024-sample blocks, one set of atomics per distinct cell",
        "cells": "k_grid_encode_bwd_cells, 16 levels: as merge, four cells per wave",
        "carry": "... + x-neighbour carry: a vertex two cells of the block share is written once"}


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("cnc::", "")


out = [f"# Headline backward call: every scatter route, time and atomic requests ({TAG})", "",
       "`tools/collect_routes.sh` + `tools/summarise_routes.py`: one 2^20-sample chunk from the middle of the bench frame, 16 levels,",
       "F = 8, STE; per route one untraced process (HIP events, median of 20 calls), one `rocprofv3 --kernel-trace --stats` process and one",
       "`rocprofv3 --pmc TCC_ATOMIC_sum` process (counters only).  Requests are per call (all kernels of the route summed).", "",
       "| route | what | ms per call | kernels (avg µs each) | atomic requests per call |", "|---|---|---|---|---|"]
for r in routes:
    t = json.loads(open(os.path.join(src, f"time_{r}.json")).read().strip().splitlines()[-1])
    ks = []
    for row in csv.DictReader(open(os.path.join(src, f"stats_{r}", "s_kernel_stats.csv"))):
        n = short(row["Name"])
        if (n.startswith("k_grid_encode_bwd") or n.startswith("k_bwd_")) and int(row["Calls"]) > 1:
            ks.append(f"{n} {float(row['AverageNs']) / 1e3:.0f}")
    per, name = defaultdict(float), {}
    for f in glob.glob(os.path.join(src, f"pmc_{r}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == "TCC_ATOMIC_sum":
                per[row["Dispatch_Id"]] += float(row["Counter_Value"])
                name[row["Dispatch_Id"]] = short(row["Kernel_Name"])
    tot, cnt = defaultdict(float), defaultdict(int)
    for d, v in per.items():
        tot[name[d]] += v
        cnt[name[d]] += 1
    # calls of the route: 24 (3 warm-up + 20 timed + 1 compared); the reference `runs` call at the end adds one dispatch of
    # k_grid_encode_bwd<3,8,false,true> to every route but `runs` itself
    calls = 24
    req = 0.0
    for n in tot:
        if not (n.startswith("k_grid_encode_bwd") or n.startswith("k_bwd_")):
            continue
        if r != "runs" and n.startswith("k_grid_encode_bwd<") :
            continue
        req += tot[n] / (calls + (1 if r == "runs" else 0))
    out.append(f"| {r} | {what[r]} | {t['ms_per_call']:.3f} | {'; '.join(ks)} | {req / 1e6:.2f} M |")
out += ["", "## Distinct cells per block size (the same chunk, per level)", "", "```"]
out += [l.rstrip() for l in open(os.path.join(src, "count.txt")) if l.startswith("level") or l.startswith("coarse")]
out += ["```"]
extra = os.path.join(src, "coarse10.txt")
if os.path.exists(extra):
    out += ["", "## The ten coarse levels alone (what the product route gives the merge kernel), per route", "", "```"]
    out += [l.rstrip() for l in open(extra) if l.startswith("{")]
    out += ["```"]
out += ["", "## What this says about a two-stage merge (a block carrying its cell table across 2 / 4 blocks)", "",
        "* The count above is the ceiling of what carrying buys on the ten coarse levels: 1.19 M distinct cells per call at 1,024 samples,",
        "  0.84 M at 2,048, 0.67 M at 4,096 — at the merge kernel's 5.6 requests per cell, 6.6 M -> 4.7 M -> 3.8 M atomic requests.",
        "* The merge kernel is not bound by them: alone it takes 0.47 ms (bench line, `bwd_coarse_levels ... alone`), 6.6 M requests at the",
        "  memory side's 21 G/s are 0.32 ms and overlap with its vector work; with the atomics compiled out it ran 10 % faster",
        "  (docs/engineering_log.md 4.2b).  Inside the product call it runs next to the binned half, whose 34 M plain requests keep the",
        "  memory-side units 0.97-1.06 busy (bench line, `memory_side_unit`): 2.8 M fewer atomic requests are 7 % of that load — a",
        "  projected 0.96 -> 0.90-0.92 ms per call at 4,096 samples, short of the 0.88 asked for.",
        "* What carrying costs: the kernel keeps SAMPLES in LDS (16 B weights + 32 B gradient row + 16 B key/record = 64 B each; 75 KB per",
        "  1,024 with the tables, two blocks per CU) and sums a cell in registers while walking its chain.  4,096 samples are 256 KB",
        "  (LDS is 160 KB); 2,048 are 128 KB = one block per CU, and every phase of the kernel ends in a barrier — one resident block has",
        "  nothing to overlap them with (the 1,024-thread form already lost to 512 at two blocks per CU until the per-cell records were",
        "  packed).  Carrying ACCUMULATORS instead (8 corners x 8 features x 4 B = 256 B per live cell, evict on conflict) needs the",
        "  adds to go through LDS atomics, which run at 0.3 lanes per clock and CU on this part (profiles/r06_step_backward_calls.md:",
        "  the row-keyed LDS kernel was 5x slower than the global-atomic one it replaced).",
        "* The lanes-of-four-cells kernel built this round for the training step's masked calls, on the same ten levels: 0.62 ms (0.72",
        "  with the carry) against the merge kernel's 0.49; on all 16 levels it is the fastest single kernel (2.05 ms against 2.23 /",
        "  2.43) but the binned route (1.05 ms for the whole call) is what the fine levels run on.",
        "", "So the headline call stays as it was: 0.96 ms, 7.0-7.5 M atomic requests."]
open(os.path.join(ROOT, "profiles", f"{TAG}_headline_backward_routes.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
