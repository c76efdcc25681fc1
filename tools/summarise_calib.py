"""gpurun_out/calib (tools/collect_calib.sh) -> profiles/r03_counter_calibration.md + profiles/counter_calibration.json.

For every probe kernel of tools/fetch_calib.hip: the bytes it is KNOWN to move against what FETCH_SIZE / WRITE_SIZE
(KB) report, and the raw L2 -> fabric request counters behind them.  The factors (known / reported) per access class
are what bench.py and tools/summarise_pmc_r03.py apply instead of a blanket 2x."""
import csv, glob, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "calib")
P = os.path.join(ROOT, "profiles")


def per_kernel(sub):
    """kernel -> counter -> average per dispatch"""
    per_dispatch, name_of = defaultdict(float), {}
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (f, r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            name_of[key] = r["Kernel_Name"]
    tot, cnt = defaultdict(float), defaultdict(int)
    for (f, d, c), v in per_dispatch.items():
        n = re.sub(r"\(.*", "", name_of[(f, d, c)]).replace("void ", "")
        tot[(n, c)] += v
        cnt[(n, c)] += 1
    out = defaultdict(dict)
    for (n, c), v in tot.items():
        out[n][c] = v / cnt[(n, c)]
    return out


timing = [json.loads(l) for l in open(os.path.join(src, "timing.jsonl")) if l.startswith("{")]
ctr = defaultdict(dict)
for sub in ("fetch", "write", "rdreq", "wrreq"):
    for n, d in per_kernel(sub).items():
        ctr[n].update(d)

# what each probe is known to read / write (bytes per launch): (reads, writes, reads at 64-byte sector granularity)
rows = []
for t in timing:
    label = t["kernel"]
    kname = label.split("/")[0]
    c = ctr.get(kname, {})
    kb = t["known_bytes"]
    if "stream_read" in label:
        rd, wr, rd64 = kb, 0, kb
    elif "stream_write" in label:
        rd, wr, rd64 = 0, kb, 0
    elif "gather32" in label:
        rd, wr, rd64 = kb, 0, kb * 2          # a 32-byte row alone in its 64-byte sector
    elif "scatter16" in label:
        rd, wr, rd64 = 0, kb, 0
    else:                                     # rmw32: half read, half written
        rd, wr, rd64 = kb / 2, kb / 2, kb / 2
    f_b = c.get("FETCH_SIZE", float("nan")) * 1024
    w_b = c.get("WRITE_SIZE", float("nan")) * 1024
    rows.append(dict(label=label, ms=t["ms"], GBps=t["GBps"], known_read=rd, known_read_sectors64=rd64, known_write=wr,
                     FETCH_SIZE_bytes=f_b, WRITE_SIZE_bytes=w_b, rdreq=c.get("TCC_EA0_RDREQ_sum"), rdreq_32B=c.get("TCC_EA0_RDREQ_32B_sum"),
                     bubble=c.get("TCC_BUBBLE_sum"), wrreq=c.get("TCC_EA0_WRREQ_sum"), wrreq_64B=c.get("TCC_EA0_WRREQ_64B_sum"),
                     fetch_factor=(rd / f_b) if rd and f_b == f_b and f_b > 0 else None,
                     fetch_factor_sectors64=(rd64 / f_b) if rd64 and f_b == f_b and f_b > 0 else None,
                     write_factor=(wr / w_b) if wr and w_b == w_b and w_b > 0 else None))
json.dump(rows, open(os.path.join(P, "counter_calibration.json"), "w"), indent=1)
g = lambda v, f="{:.3f}": "-" if v is None or v != v else f.format(v)
with open(os.path.join(P, "r03_counter_calibration.md"), "w") as fh:
    fh.write("# rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 against known byte counts (tools/fetch_calib.hip)\n\n"
             "`known` = bytes the kernel is written to move per launch; `reported` = counter x 1024; factor = known / reported.\n"
             "`sector` factor = (64-byte sectors touched) / reported, for gathers of a 32-byte row.\n\n"
             "| probe | ms | GB/s (known) | known read | FETCH_SIZE reported | factor | sector factor | known write | WRITE_SIZE reported | factor | RDREQ | RDREQ_32B | BUBBLE | WRREQ | WRREQ_64B |\n"
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        fh.write(f"| {r['label']} | {r['ms']:.3f} | {r['GBps']:.0f} | {r['known_read']/1e6:.1f} MB | {g(r['FETCH_SIZE_bytes']/1e6 if r['FETCH_SIZE_bytes']==r['FETCH_SIZE_bytes'] else None, '{:.1f} MB')} | "
                 f"{g(r['fetch_factor'])} | {g(r['fetch_factor_sectors64'])} | {r['known_write']/1e6:.1f} MB | "
                 f"{g(r['WRITE_SIZE_bytes']/1e6 if r['WRITE_SIZE_bytes']==r['WRITE_SIZE_bytes'] else None, '{:.1f} MB')} | {g(r['write_factor'])} | "
                 f"{g(r['rdreq'], '{:.0f}')} | {g(r['rdreq_32B'], '{:.0f}')} | {g(r['bubble'], '{:.0f}')} | {g(r['wrreq'], '{:.0f}')} | {g(r['wrreq_64B'], '{:.0f}')} |\n")
print(open(os.path.join(P, "r03_counter_calibration.md")).read())
