"""gpurun_out/<tag> (tools/collect_profiles.sh <tag>) -> profiles/<tag>_pmc_hbm_traffic.csv, profiles/traffic.json and
copies of the kernel-stats CSVs.      python tools/summarise_pmc.py r04

Counter arithmetic (profiles/r03_counter_calibration.md, tools/fetch_calib.hip):
  * FETCH_SIZE (KB) = L2 -> fabric read requests x 64 B.  A coalesced 16 B/lane stream moves 128 B per request (reported
    = 1/2 of the known bytes); a gather of one 32-byte row is ONE request whatever it carries (reported = the 64-byte
    sectors touched).  So:  read bytes = 64 B x requests + 64 B x (requests of the kernel's STREAMED reads), the second
    term from the byte count the kernel is known to stream (items, table slabs, inputs, upstream gradient).
  * WRITE_SIZE (KB) is exact for 64-byte write requests (streaming writes: factor 1.000) and charges 32 B for a
    partial one (a lone 16-byte store: factor 0.5, i.e. it over-counts the useful bytes 2x) — used as reported.
traffic.json carries, per entry point, the calibrated bytes, both bounds (every read request 64 B / 128 B), the request
counts, and the git blob hashes of the kernel sources the numbers were measured on (bench.py flags a mismatch)."""
import csv, glob, hashlib, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", TAG)
P = os.path.join(ROOT, "profiles")
KERNEL_SOURCES = ["cnc_amd/csrc/grid_encode.hip", "cnc_amd/csrc/grid_encode_merge.hip", "cnc_amd/csrc/grid_encode_binned.hip",
                  "cnc_amd/csrc/grid_encode_overlap.hip", "cnc_amd/csrc/encoder_common.hpp", "cnc_amd/csrc/common.hpp",
                  "cnc_amd/csrc/march.hip"]


def blob_hash(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


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
rdreq = per_kernel(os.path.join(src, "pmc_req"), "TCC_EA0_RDREQ_sum")
wrreq = per_kernel(os.path.join(src, "pmc_req"), "TCC_EA0_WRREQ_sum")
atom = per_kernel(os.path.join(src, "pmc_atomic"), "TCC_ATOMIC_sum")
names = sorted(n for n in set(fetch) | set(write) | set(atom) if n.startswith("cnc::"))
get = lambda d, n: d.get(n, (0.0, 0))[0]
cnt = {n: max(fetch.get(n, (0, 0))[1], write.get(n, (0, 0))[1], atom.get(n, (0, 0))[1]) for n in names}
with open(os.path.join(P, f"{TAG}_pmc_hbm_traffic.csv"), "w") as fh:
    fh.write("kernel,FETCH_SIZE_avg_KB_raw,WRITE_SIZE_avg_KB,read_requests_avg,write_requests_avg,TCC_ATOMIC_sum_avg,dispatches\n")
    for n in names:
        fh.write(f'"{n}",{get(fetch, n):.1f},{get(write, n):.1f},{get(rdreq, n):.0f},{get(wrreq, n):.0f},{get(atom, n):.0f},{cnt[n]}\n')

L, F, L_BINNED = 16, 8, 6        # levels of the bench grid on the binned path (gridencoder_backend.plan_binned_levels)
# samples per encoder call, averaged over the frame's calls like the counters below (the last call of a frame is partial)
N_CHUNK = 1 << 20
_bj = os.path.join(src, "bench.json")
if os.path.exists(_bj) and os.path.getsize(_bj) > 0:
    try:
        N_CHUNK = int(round(json.loads([l for l in open(_bj) if l.startswith("{")][0])["roofline"]["samples_per_launch"]))
    except Exception:
        pass


def call_total(d, pat, calls):
    return sum(get(d, n) * cnt[n] for n in names if re.search(pat, n)) / calls


calls = max(sum(cnt[n] for n in names if re.search(r"k_grid_encode_bwd_merge<", n)), 1)
# the bench frame's forward is the 3-D instantiation only: the `field` block of bench.py also dispatches the 2-D planes' kernel
FWD_PAT = r"k_grid_encode_fwd_bits<3u"
fwd_calls = max(sum(cnt[n] for n in names if re.search(FWD_PAT, n)), 1)


def entry(pats, n_calls, streamed_read_bytes, note):
    """calibrated bytes of one entry point = 64 B x read requests + 64 B x (requests that carried a 128-byte stream)
    + WRITE_SIZE; bounds with every read request at 64 B / 128 B."""
    pat = "|".join(pats)
    f_b = call_total(fetch, pat, n_calls) * 1024
    w_b = call_total(write, pat, n_calls) * 1024
    rd = call_total(rdreq, pat, n_calls)
    wr = call_total(wrreq, pat, n_calls)
    at = call_total(atom, pat, n_calls)
    streamed_req_bytes = min(streamed_read_bytes / 2.0, f_b)      # a 128-byte stream request is tallied as 64 B
    return {"bytes": f_b + streamed_req_bytes + w_b, "bytes_min(read requests x 64 B)": f_b + w_b,
            "bytes_max(read requests x 128 B)": 2 * f_b + w_b, "FETCH_SIZE_bytes_raw": f_b, "WRITE_SIZE_bytes": w_b,
            "read_requests": rd or f_b / 64.0, "write_requests": wr, "atomic_requests": at,
            "streamed_read_bytes_known": streamed_read_bytes, "note": note}


# ---- SURVEY 8(d) Input B (tools/bench_input_b.py under the same counters, its own passes: pmc_b_*) ----
def input_b_entries():
    fb = per_kernel(os.path.join(src, "pmc_b_fetch"), "FETCH_SIZE")
    wb = per_kernel(os.path.join(src, "pmc_b_write"), "WRITE_SIZE")
    rb = per_kernel(os.path.join(src, "pmc_b_req"), "TCC_EA0_RDREQ_sum")
    wrb = per_kernel(os.path.join(src, "pmc_b_req"), "TCC_EA0_WRREQ_sum")
    ab = per_kernel(os.path.join(src, "pmc_b_atomic"), "TCC_ATOMIC_sum")
    if not fb:
        return {}
    nb = sorted(n for n in set(fb) | set(wb) | set(ab) if n.startswith("cnc::"))
    cb = {n: max(fb.get(n, (0, 0))[1], wb.get(n, (0, 0))[1], ab.get(n, (0, 0))[1]) for n in nb}
    with open(os.path.join(P, f"{TAG}_pmc_hbm_traffic_input_B.csv"), "w") as fh:
        fh.write("kernel,FETCH_SIZE_avg_KB_raw,WRITE_SIZE_avg_KB,read_requests_avg,write_requests_avg,TCC_ATOMIC_sum_avg,dispatches\n")
        for n in nb:
            fh.write(f'"{n}",{get(fb, n):.1f},{get(wb, n):.1f},{get(rb, n):.0f},{get(wrb, n):.0f},{get(ab, n):.0f},{cb[n]}\n')
    N_B, LEVELS = 1 << 18, 24

    def tot(d, pats, calls):
        return sum(get(d, n) * cb[n] for n in nb if any(re.search(p_, n) for p_ in pats)) / calls

    out = {}
    for Fb in (8, 2):
        for tag, kern, streamed in (("fwd", "k_grid_encode_fwd_bits", N_B * 36),
                                    ("bwd", "k_grid_encode_bwd", N_B * 36 + N_B * LEVELS * Fb * 4)):
            pats = [rf"{kern}<3u, {Fb}u", rf"{kern}<2u, {Fb}u"]
            calls = max(sum(cb[n] for n in nb if re.search(rf"{kern}<3u, {Fb}u", n)), 1)
            f_b, w_b = tot(fb, pats, calls) * 1024, tot(wb, pats, calls) * 1024
            sreq = min(streamed / 2.0, f_b)
            out[f"input_B_{tag}_F{Fb}"] = {
                "bytes": f_b + sreq + w_b, "bytes_min(read requests x 64 B)": f_b + w_b, "bytes_max(read requests x 128 B)": 2 * f_b + w_b,
                "FETCH_SIZE_bytes_raw": f_b, "WRITE_SIZE_bytes": w_b, "read_requests": tot(rb, pats, calls) or f_b / 64.0,
                "write_requests": tot(wrb, pats, calls), "atomic_requests": tot(ab, pats, calls), "streamed_read_bytes_known": streamed,
                "call_sets": calls,
                "note": "one call SET = the four encoders (12 x 3-D + 3 planes x 4 levels) on 2^18 marched samples, as the training "
                        "step's render pass issues them (tools/bench_input_b.py)"}
    return out


items = 4 * N_CHUNK * L_BINNED * 16                       # one 16-byte item per (sample, binned level, corner pair)
slabs = L_BINNED * (1 << 19) * F * 4                      # the owners read every table slab of the binned levels once
traffic = {
    "grid_encode_forward": entry([FWD_PAT], fwd_calls, N_CHUNK * 12,
                                 "reads: the 12-byte points streamed, byte gathers from the 6 MB sign plane (L2 / Infinity-Cache hits "
                                 "mostly); writes: the [L, N, F] output stream"),
    "grid_encode_backward": entry([r"k_grid_encode_bwd_merge<", r"k_bwd_bin", r"k_bwd_owner"], calls,
                                  N_CHUNK * 12 * 2 + N_CHUNK * (L - L_BINNED) * F * 4 + items + slabs,
                                  "streamed: points (both halves), the coarse levels' gradient rows, the items, the table slabs; the "
                                  "rest of the read requests are the owners' gathers of 32-byte gradient rows (one request each)"),
    "k_bwd_bin+k_bwd_owner": entry([r"k_bwd_bin", r"k_bwd_owner"], calls, N_CHUNK * 12 + items + slabs,
                                   "finest levels alone"),
    "k_grid_encode_bwd_merge": entry([r"k_grid_encode_bwd_merge<"], calls, N_CHUNK * 12 + N_CHUNK * (L - L_BINNED) * F * 4,
                                     "coarse levels alone"),
    # cnc_march_samples = the count pass with positions requested (<0, 32, true>) + the resumed fill (<2, ...>); the
    # <0, 32, false> / <1, ...> dispatches are the drop-in traverse_grids probe of bench.py's kernel table
    "march_samples(count+fill)": entry([r"k_traverse<0, \d+, true>", r"k_traverse<2"], max(cnt.get(next((n for n in names if "k_traverse<2" in n), ""), 1), 1),
                                       2 * 640000 * 24, "per 640k-ray frame; the fill pass also writes the positions (12 B / sample)"),
    **input_b_entries(),
    "_sources": {p: blob_hash(os.path.join(ROOT, p)) for p in KERNEL_SOURCES if os.path.exists(os.path.join(ROOT, p))},
    "_fabric_request_rate_peak_G_per_s": 50.0,
    "_samples_per_call": N_CHUNK,
    "_note": "per encoder call of the bench frame (_samples_per_call samples on average); separate rocprofv3 --pmc passes of `python bench.py --steps 2 "
             "--warmup 1 --no-cpu-baseline --no-train-step` (tools/collect_profiles.sh, profiles/" + TAG + "_pmc_hbm_traffic.csv); counter "
             "arithmetic per profiles/r03_counter_calibration.md: FETCH_SIZE = read requests x 64 B, a streamed request carries 128 B, a "
             "gathered 32-byte row is one request; the fabric sustains ~50 G requests/s (streaming 46.9, gathers 51.5: "
             "tools/fetch_calib.hip)",
}
json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)
for sub, dst in (("stats", f"{TAG}_bench_kernel_stats.csv"), ("stats_no_overlap", f"{TAG}_bench_kernel_stats_no_overlap.csv"),
                 ("stats_train", f"{TAG}_train_step_kernel_stats.csv")):
    hits = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(P, dst))
for f, dst in (("bench.json", f"{TAG}_bench.json"), ("bench_no_overlap.json", f"{TAG}_bench_no_overlap.json")):
    if os.path.exists(os.path.join(src, f)) and os.path.getsize(os.path.join(src, f)) > 0:
        shutil.copy(os.path.join(src, f), os.path.join(P, dst))
if os.path.exists(os.path.join(src, "train.log")):
    keep = [l for l in open(os.path.join(src, "train.log")) if l.startswith("train step") or l.startswith("setup")]
    open(os.path.join(P, f"{TAG}_train_step.log"), "w").write(
        "# tools/bench_train.py --no-profile under rocprofv3 --kernel-trace --stats (tracing overhead included;\n"
        "# untraced: tools/profile_train_step.py / bench.py train_step)\n" + "".join(keep))
hits = glob.glob(os.path.join(src, "stats_eval", "**", "*kernel_stats.csv"), recursive=True)
if hits:
    shutil.copy(hits[0], os.path.join(P, f"{TAG}_eval_render_kernel_stats.csv"))


def _grep(path, pat):
    return [l.rstrip() for l in open(path)] if os.path.exists(path) and pat is None else \
        ([l.rstrip() for l in open(path) if re.search(pat, l)] if os.path.exists(path) else [])


# the training step's phase timeline + where the library launches come from
ph = _grep(os.path.join(src, "step_phases.txt"), r"^boundary|^\w+(:begin)? +\w{3} +[0-9.]+ +[0-9.]+")
ab = _grep(os.path.join(src, "aten_by_range.txt"), None)
if ph:
    with open(os.path.join(P, f"{TAG}_train_step_phases.md"), "w") as fh:
        fh.write(f"# {TAG} — where a training step's wall time goes (full model, F = 8, one MI355X)\n\n"
                 "`python tools/step_phases.py`: host clock and a HIP event at every phase boundary of `Trainer.train_step`, both\n"
                 "host threads (`Mai` = main: fetch, occupancy, render, optimisers; `cnc` = the context thread), no profiler;\n"
                 "host ms / GPU ms since the step's start, mean over the non-refresh steps.  host > gpu at a boundary: the GPU\n"
                 "had finished that work before the host returned (host-bound there); gpu > host: the GPU is behind.\n\n```\n"
                 + "\n".join(ph) + "\n```\n\n" +
                 "\n".join(_grep(os.path.join(src, "train_untraced.log"), r"^train step")) + "\n")
        if ab:
            i = next((k for k, l in enumerate(ab) if l.startswith("library kernels")), 0)
            fh.write("\n## Library (ATen / rocPRIM / copy) launches of one step by call site (`tools/aten_by_range.py`, sequential schedule)\n\n```\n"
                     + "\n".join(ab[i:i + 45]) + "\n```\n")

# MFMA busy of the fused field kernels (and of the chain's GEMMs beside them)
rows = []
for sub, label in (("pmc_field", "default (fp16 three-product MFMA)"), ("pmc_field_f32", "CNC_FUSED_FIELD_MFMA=f32 (exact fp32 MFMA)")):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if not ("k_field_fused" in k or k.startswith("Cijk") or "fwd_bits" in k):
            continue
        n = len(d.get("GRBM_GUI_ACTIVE", []))
        if not n:
            continue
        mean = lambda c: sum(d.get(c, [0])) / max(len(d.get(c, [0])), 1)
        cyc = mean("GRBM_GUI_ACTIVE") / 8.0                      # per XCD: the kernel's duration in shader cycles
        rows.append((label, k[:70], n, mean("SQ_INSTS_MFMA"), mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("SQ_INSTS_VALU"), cyc,
                     mean("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / max(cyc, 1), mean("SQ_INSTS_VALU") * 4.0 / 1024.0 / max(cyc, 1)))
if rows:
    with open(os.path.join(P, f"{TAG}_mfma_utilisation.md"), "w") as fh:
        fh.write(f"# {TAG} — matrix-pipe and vector-pipe occupancy of the gradient-free field (MI355X, N = 2^20, F = 8)\n\n"
                 "`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE`\n"
                 "around `tools/bench_field.py` (fused kernel and the chain it replaces; `tools/collect_profiles.sh`).  duration =\n"
                 "GRBM_GUI_ACTIVE / 8 XCDs (shader cycles, profiled run); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs /\n"
                 "duration; vector issue = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / duration (MFMA instructions included).  The f32\n"
                 "MFMA runs at the vector rate and shares its issue with the gather (busy + issue ~ the kernel); the fp16 form is\n"
                 "a sixth of the matrix cycles on the real matrix pipe (DESIGN 4.5).\n\n"
                 "| run | kernel | dispatches | MFMA instr | MFMA busy cycles | VALU instr | duration (cycles) | MFMA busy | vector issue |\n|---|---|---|---|---|---|---|---|---|\n")
        for r in sorted(rows, key=lambda t: (t[0], -t[6])):
            fh.write(f"| {r[0]} | `{r[1]}` | {r[2]} | {r[3]:.3g} | {r[4]:.3g} | {r[5]:.3g} | {r[6]:.3g} | {r[7] * 100:.1f} % | {r[8] * 100:.1f} % |\n")
        fh.write("\nWall clock of the same calls, untraced (`tools/bench_field.py`):\n\n```\n" +
                 "\n".join(_grep(os.path.join(src, "field_untraced.log"), r"^(fused|chain)")) + "\n```\n")
ev = _grep(os.path.join(src, "eval_untraced.log"), r"^eval render")
if ev:
    open(os.path.join(P, f"{TAG}_eval_render.log"), "w").write("# tools/bench_eval.py, untraced (200 training steps of the full model, then 800x800)\n" + "\n".join(ev) + "\n")
print(open(os.path.join(P, f"{TAG}_pmc_hbm_traffic.csv")).read())
print(json.dumps({k: v for k, v in traffic.items() if not k.startswith("_")}, indent=1)[:3000])
