"""What the box's HBM does for a pure read, a pure write and a copy of ~2.5 GB (torch's own elementwise kernels: a yardstick
next to the in-library stream copy, not a product path):  python scripts/bw_ceilings.py"""
import json
import torch

n = 640 * 1024 * 1024  # floats: 2.5 GiB
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)


def t(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ms_r = t(lambda: x.sum())
ms_w = t(lambda: y.fill_(1.0))
ms_c = t(lambda: y.copy_(x))
gb = n * 4 / 1e9
print(json.dumps({"gb": round(gb, 3), "read_sum_ms": round(ms_r, 4), "read_tb_s": round(gb / ms_r, 3), "fill_ms": round(ms_w, 4),
                  "write_tb_s": round(gb / ms_w, 3), "copy_ms": round(ms_c, 4), "copy_tb_s": round(2 * gb / ms_c, 3)}))
