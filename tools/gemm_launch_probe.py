"""Scratch: host-side cost of a hipBLASLt GEMM launch through torch when the row count is new every call
(the sample count of a training step) against a repeated row count."""
import time, torch
dev = torch.device("cuda:0")
W = torch.randn(160, 256, device=dev); b = torch.randn(160, device=dev)
def run(ms, what):
    xs = [torch.randn(m, 256, device=dev) for m in ms]
    torch.cuda.synchronize()
    t = []
    for x in xs:
        t0 = time.perf_counter()
        if what == "addmm_act": torch._addmm_activation(b, x, W.t())
        elif what == "linear": torch.nn.functional.linear(x, W, b)
        elif what == "mm": x @ W.t()
        elif what == "bmm":
            s = 256; m = x.shape[0] // s
            torch.bmm(x[:m * s].view(s, m, -1).transpose(1, 2), x[:m * s].view(s, m, -1))
        t.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    t = sorted(t)
    return 1e6 * t[len(t) // 2]
same = [262144] * 40
new = [250000 + 37 * i for i in range(40)]
bucket = [(250000 + 997 * i + 4095) // 4096 * 4096 for i in range(40)]
for what in ("addmm_act", "linear", "mm", "bmm"):
    run(same, what)
    print(f"{what:10s} median host us: same M {run(same, what):7.1f}   new M every call {run(new, what):7.1f}   bucketed (4096) {run(bucket, what):7.1f}")
