"""SURVEY 8(d) Input B alone (bench.py's `input_B_entries`: the four encoders of the reference composition on 2^18 marched
samples, F = 8 and F = 2, forward and backward as the training step's render pass calls them) — the command the PMC
passes of tools/collect_profiles.sh profile.  Prints the per-call-set times.   --only fwd8|bwd8|fwd2|bwd2 : one of them"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
w = bench.build_workload(dev, 0)
box = {}
bench.march_frame(w, box)                # one frame's march only: no encoder kernel of the bench frame in this process
torch.cuda.synchronize()
k, r = bench.input_B_entries(dev, bench.probe_chunk_of(box["ex"]["positions"]), {})
print(json.dumps({n: round(v["avg_ms"], 4) for n, v in k.items()}))
print(json.dumps({n: {q: v[q] for q in ("avg_launch_ms", "compulsory_bytes", "touched_table_rows", "achieved_algorithmic")}
                  for n, v in r.items() if isinstance(v, dict)}))
