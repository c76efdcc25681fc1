"""Which package lines issue the ATen ops of a steady-state training step (forward side, sequential schedule): a
TorchDispatchMode counts every op that reaches the dispatcher against the innermost cnc_amd/ frame on the Python stack.
    python tools/ops_by_line.py"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CNC_CTX_THREAD", "0")
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from cnc_amd.trainer import TrainConfig, Trainer

cfg = TrainConfig(n_features=8, sample_num=150000, max_steps=400, image_size=400, out_dir="/tmp/bits")
tr = Trainer(cfg, device=torch.device("cuda:0"))
for step in range(243):
    tr.train_step(step, want_stats=False)
torch.cuda.synchronize()
SKIP = ("aten.view", "aten.detach", "aten.alias", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.unsqueeze", "aten.squeeze",
        "aten.select", "aten.slice", "aten.expand", "aten.permute", "aten.as_strided", "aten.reshape", "aten.unbind", "aten.split",
        "aten.empty", "aten.narrow", "aten._local_scalar_dense", "aten.is_", "aten.sym_", "aten.lift_fresh", "aten.result_type")
acc = collections.Counter()
ops = collections.defaultdict(collections.Counter)


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "?"
            for fs in reversed(traceback.extract_stack(limit=40)):
                if "cnc_amd/" in fs.filename and not fs.filename.endswith("_lib.py"):
                    where = f"{fs.filename.split('cnc_amd/')[-1]}:{fs.lineno} {fs.name}"
                    break
            acc[where] += 1
            ops[where][name.replace("aten.", "")] += 1
        return func(*args, **(kwargs or {}))


with Count():
    tr.train_step(243, want_stats=False)
torch.cuda.synchronize()
print("dispatcher ops of the forward side of one step (views / empties not counted):", sum(acc.values()))
for where, n in acc.most_common(70):
    print(f"{n:4d}  {where:60s} " + " ".join(f"{k}x{v}" if v > 1 else k for k, v in ops[where].most_common(6)))
