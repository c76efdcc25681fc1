"""gpurun_out/mfma_r03 (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python tools/mlp_pmc_r03.py)
-> profiles/r03_mfma_utilisation.md.  utilisation = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), as r01."""
import csv, glob, os, re, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "mfma_r03")
busy, gui, n = defaultdict(float), defaultdict(float), defaultdict(int)
seen = set()
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[k] += float(r["Counter_Value"])
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            gui[k] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen:
            seen.add((r["Dispatch_Id"], k)); n[k] += 1
dur = defaultdict(float)
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
rows = [(k, busy[k], gui[k], n[k], dur.get(k, 0.0)) for k in busy if busy[k] > 0]
rows.sort(key=lambda t: -t[1])
out = os.path.join(ROOT, "profiles", "r03_mfma_utilisation.md")
with open(out, "w") as fh:
    fh.write("# r03 — MFMA utilisation of the radiance-field MLP work (MI355X, fp32 MFMA peak 157.3 TFLOP/s)\n\n"
             "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python tools/mlp_pmc_r03.py`: three training\n"
             "steps of the two networks at N = 2^18 (forward with bias+ReLU epilogue, dX, split-K dW; cnc_amd/mlp.py on hipBLASLt),\n"
             "three evaluations of each network at N = 2^20 by the hand-written fused kernel and by the library chain.\n"
             "utilisation = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); one `v_mfma_f32_16x16x4_f32` = 32 cycles = 1024 MAC,\n"
             "so useful TFLOP/s = busy cycles x 64 FLOP / duration.\n\n"
             "| kernel | dispatches | MFMA busy cycles | duration (µs, profiled, all dispatches) | MFMA utilisation | TFLOP/s issued |\n|---|---|---|---|---|---|\n")
    for k, b, g, c, d in rows:
        util = b / (g / 8 * 1024) if g else float("nan")
        tf = b * 64 / (d * 1e-6) / 1e12 if d else float("nan")
        fh.write(f"| `{k[:120]}` | {c} | {b:.3e} | {d:.0f} | {util * 100:.1f} % | {tf:.1f} |\n")
print(open(out).read())
