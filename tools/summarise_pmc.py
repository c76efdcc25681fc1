"""Per-kernel HBM traffic from the two rocprofv3 --pmc passes made by tools/collect_profiles.sh.
FETCH_SIZE / WRITE_SIZE are in KB; FETCH is doubled on gfx950 (MI355X_MICROARCH.md, HBM section).
Writes profiles/r01_pmc_hbm_traffic.csv and profiles/traffic.json (bytes per encoder call)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r01")


def per_kernel(counter_dir, counter):
    per_dispatch = defaultdict(float)
    name_of = {}
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
rows = []
for n in sorted(set(fetch) | set(write)):
    if not n.startswith("cnc::"):
        continue
    f, w = fetch.get(n, (0, 0))[0], write.get(n, (0, 0))[0]
    rows.append((n, f, w, (2 * f + w) * 1024, fetch.get(n, (0, 0))[1]))
with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.csv"), "w") as fh:
    fh.write("kernel,FETCH_SIZE_avg_KB_raw,WRITE_SIZE_avg_KB,hbm_bytes_per_launch(2*FETCH+WRITE)*1024,dispatches\n")
    for n, f, w, b, c in rows:
        fh.write(f'"{n}",{f:.1f},{w:.1f},{b:.0f},{c}\n')
by = {n: b for n, f, w, b, c in rows}
cnt = {n: c for n, f, w, b, c in rows}
pick = lambda pat: sum(v for k, v in by.items() if re.search(pat, k))
# bytes per CALL for kernels that are dispatched more than once per call (the binned levels go out as
# two groups): total bytes of the kernel / number of calls (= dispatches of the coarse kernel)
total = lambda pat: sum(by[k] * cnt[k] for k in by if re.search(pat, k))
calls = max(sum(cnt[k] for k in by if re.search(r"k_grid_encode_bwd(_merge)?<", k)), 1)
traffic = {
    "grid_encode_forward": pick(r"k_grid_encode_fwd_bits"),
    # one backward call = atomic kernel (coarse levels; run-merging variant in the bench) + bin pass +
    # owner pass (finest levels)
    "grid_encode_backward": (total(r"k_grid_encode_bwd(_merge)?<") + total(r"k_bwd_bin") + total(r"k_bwd_owner")) / calls,
    "_note": "HBM bytes per encoder call on a 2^20-sample chunk = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the "
             "call's kernels, from separate rocprofv3 --pmc passes of `python bench.py --steps 2 --warmup 1 "
             "--no-cpu-baseline` (profiles/r01_pmc_hbm_traffic.csv, tools/collect_profiles.sh); FETCH doubled per "
             "the gfx950 note in MI355X_MICROARCH.md",
}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.csv")).read())
print(traffic)
