"""Which call sites launch the library (ATen / rocPRIM / copy) kernels of a steady-state training step: one profiled
step, every non-cnc, non-hipBLASLt kernel attributed to its outermost op and to the innermost named range
(record_function) around it; counts and GPU time per (site, kernel).
    CNC_CTX_THREAD=0 CNC_PROFILE_RANGES=1 python tools/aten_by_range.py [--all]"""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CNC_CTX_THREAD", "0")
os.environ.setdefault("CNC_PROFILE_RANGES", "1")
import torch
from torch.profiler import ProfilerActivity, profile

from cnc_amd.trainer import TrainConfig, Trainer

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(243):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
STACK = "--stack" in sys.argv       # also: the package line (first cnc_amd/ frame) each library launch comes from
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=STACK) as prof:
    tr.train_step(243, want_stats=False)      # not a refresh step (243 % 16 = 3)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("at::native::", "")
    n = re.sub(r"<.*", "", n)
    return n[:48]


def lib(k):
    return not (k.name.startswith("cnc::") or "cnc::" in k.name[:16] or k.name.startswith("Cijk") or "_ZN3cnc" in k.name)


def frame(e):
    x = e
    while x is not None:
        for f in (x.stack or []):
            if "cnc_amd/" in f and "_lib.py" not in f:
                return re.sub(r".*cnc_amd/", "", f)[:60]
        x = x.cpu_parent
    return "(autograd thread / no package frame)"


by_line = collections.defaultdict(lambda: [0, 0.0])
acc = collections.defaultdict(lambda: [0, 0.0])
for e in evs:
    if not e.kernels:
        continue
    chain = []
    x = e
    while x is not None:
        chain.append(x.name)
        x = x.cpu_parent
    rng = [n for n in chain if "/" in n and not n.startswith("aten::") and "::" not in n]
    node = [n for n in chain if "evaluate_function" in n or n.endswith("Backward") or "Backward" in n]
    site = (rng[0] if rng else "-") + " | " + (re.sub(r"autograd::engine::evaluate_function: ", "", node[-1])[:40] if node else chain[-1][:40])
    for k in e.kernels:
        if "--all" in sys.argv or lib(k):
            a = acc[(site, e.name[:28], short(k.name))]
            a[0] += 1
            a[1] += k.duration
            if STACK:
                b = by_line[frame(e) if site.startswith("- | aten::") or site.startswith("- | hip") else site]
                b[0] += 1
                b[1] += k.duration
tot_n = sum(a[0] for a in acc.values())
tot_t = sum(a[1] for a in acc.values())
print(f"library kernels in one step: {tot_n} launches, {tot_t / 1e3:.3f} ms")
by_site = collections.defaultdict(lambda: [0, 0.0])
for (site, op, k), a in acc.items():
    by_site[site][0] += a[0]
    by_site[site][1] += a[1]
print("---- by site")
for site, a in sorted(by_site.items(), key=lambda t: -t[1][0])[:60]:
    print(f"{a[0]:4d} {a[1]:8.0f}us  {site}")
print("---- by (site, op, kernel)")
for (site, op, k), a in sorted(acc.items(), key=lambda t: -t[1][0])[:120]:
    print(f"{a[0]:4d} {a[1]:8.0f}us  {site:70s} {op:28s} {k}")
if STACK:
    print("---- by package line (top-level ops) / site")
    for site, a in sorted(by_line.items(), key=lambda t: -t[1][0])[:80]:
        print(f"{a[0]:4d} {a[1]:8.0f}us  {site}")
