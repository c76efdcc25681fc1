"""gpurun_out/r02 (tools/collect_profiles_r02.sh) -> profiles/r02_pmc_hbm_traffic.csv, profiles/traffic.json and
copies of the kernel-stats CSVs.  FETCH_SIZE / WRITE_SIZE are in KB; FETCH is doubled on gfx950
(MI355X_MICROARCH.md, HBM section).  TCC_ATOMIC_sum = L2 atomic requests."""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02")
P = os.path.join(ROOT, "profiles")


def per_kernel(counter_dir, counter):
    per_dispatch, name_of = defaultdict(float), {}
    for f in glob.glob(os.path.join(counter_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = (f, r["Dispatch_Id"])
            per_dispatch[key] += float(r["Counter_Value"])
            name_of[key] = r["Kernel_Name"]
    tot, cnt = defaultdict(float), defaultdict(int)
    for k, v in per_dispatch.items():
        n = re.sub(r"\(.*", "", name_of[k]).replace("void ", "")
        tot[n] += v
        cnt[n] += 1
    return {n: (tot[n] / cnt[n], cnt[n]) for n in tot}


fetch = per_kernel(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
write = per_kernel(os.path.join(src, "pmc_write"), "WRITE_SIZE")
atom = per_kernel(os.path.join(src, "pmc_atomic"), "TCC_ATOMIC_sum")
rows = []
for n in sorted(set(fetch) | set(write) | set(atom)):
    if not n.startswith("cnc::"):
        continue
    f, w, a = fetch.get(n, (0, 0))[0], write.get(n, (0, 0))[0], atom.get(n, (0, 0))[0]
    rows.append((n, f, w, (2 * f + w) * 1024, a, max(fetch.get(n, (0, 0))[1], atom.get(n, (0, 0))[1])))
with open(os.path.join(P, "r02_pmc_hbm_traffic.csv"), "w") as fh:
    fh.write("kernel,FETCH_SIZE_avg_KB_raw,WRITE_SIZE_avg_KB,hbm_bytes_per_launch(2*FETCH+WRITE)*1024,TCC_ATOMIC_sum_avg,dispatches\n")
    for n, f, w, b, a, c in rows:
        fh.write(f'"{n}",{f:.1f},{w:.1f},{b:.0f},{a:.0f},{c}\n')
by = {n: b for n, f, w, b, a, c in rows}
at = {n: a for n, f, w, b, a, c in rows}
cnt = {n: c for n, f, w, b, a, c in rows}
pick = lambda pat: sum(v for k, v in by.items() if re.search(pat, k))
total = lambda d, pat: sum(d[k] * cnt[k] for k in d if re.search(pat, k))
calls = max(sum(cnt[k] for k in by if re.search(r"k_grid_encode_bwd_merge<", k)), 1)
traffic = {
    "grid_encode_forward": pick(r"k_grid_encode_fwd_bits"),
    # one backward call = merging atomic kernel (coarse levels) + bin passes + owner passes (finest levels, two groups)
    "grid_encode_backward": (total(by, r"k_grid_encode_bwd_merge<") + total(by, r"k_bwd_bin") + total(by, r"k_bwd_owner")) / calls,
    "k_bwd_bin+k_bwd_owner": (total(by, r"k_bwd_bin") + total(by, r"k_bwd_owner")) / calls,
    "k_grid_encode_bwd_merge": total(by, r"k_grid_encode_bwd_merge<") / calls,
    "k_grid_encode_bwd_merge_atomic_requests": total(at, r"k_grid_encode_bwd_merge<") / calls,
    "march_samples(count+fill)": pick(r"k_traverse<0") + pick(r"k_traverse<2"),
    "_note": "HBM bytes per call on a 2^20-sample chunk (march: per 640k-ray frame) = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed "
             "over the call's kernels; atomic requests = TCC_ATOMIC_sum; separate rocprofv3 --pmc passes of `python bench.py "
             "--steps 2 --warmup 1 --no-cpu-baseline --no-train-step` (profiles/r02_pmc_hbm_traffic.csv, "
             "tools/collect_profiles_r02.sh); FETCH doubled per the gfx950 note in MI355X_MICROARCH.md",
}
json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)
for sub, dst in (("stats", "r02_bench_kernel_stats.csv"), ("stats_no_overlap", "r02_bench_kernel_stats_no_overlap.csv"),
                 ("stats_train", "r02_train_step_kernel_stats.csv")):
    hits = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(P, dst))
for f, dst in (("bench.json", "r02_bench.json"), ("bench_no_overlap.json", "r02_bench_no_overlap.json")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(P, dst))
if os.path.exists(os.path.join(src, "train.log")):      # only the result line (the rest is rocprofv3 chatter)
    keep = [l for l in open(os.path.join(src, "train.log")) if l.startswith("train step") or l.startswith("setup")]
    open(os.path.join(P, "r02_train_step.log"), "w").write(
        "# tools/bench_train.py --no-profile under rocprofv3 --kernel-trace --stats (tracing overhead included;\n"
        "# untraced: tools/profile_train_step.py / bench.py train_step)\n" + "".join(keep))
print(open(os.path.join(P, "r02_pmc_hbm_traffic.csv")).read())
print(json.dumps(traffic, indent=1))
