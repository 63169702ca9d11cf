"""What the HBM of this box gives plain streams (torch kernels, HIP events): fill (write only), copy (read + write), read-mostly reduce.
Sizes as march_records_kernel's streams at npt-flange@1600: 245 MB written, 136 MB read."""
import torch
dev = torch.device("cuda:0")
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
w = torch.empty(245_000_000 // 4, dtype=torch.float32, device=dev)
r = torch.empty(136_000_000 // 4, dtype=torch.float32, device=dev).normal_()
big = torch.empty(1_000_000_000 // 4, dtype=torch.float32, device=dev)
big2 = torch.empty_like(big)
ms = t(lambda: w.fill_(1.0)); print(f"fill 245 MB: {ms*1e3:.1f} us = {0.245/ms:.2f} TB/s write")
ms = t(lambda: big.fill_(1.0)); print(f"fill 1 GB: {ms*1e3:.1f} us = {1.0/ms:.2f} TB/s write")
ms = t(lambda: big2.copy_(big)); print(f"copy 1 GB: {ms*1e3:.1f} us = {2.0/ms:.2f} TB/s read+write")
ms = t(lambda: r.sum()); print(f"sum 136 MB: {ms*1e3:.1f} us = {0.136/ms:.2f} TB/s read")
ms = t(lambda: big.sum()); print(f"sum 1 GB: {ms*1e3:.1f} us = {1.0/ms:.2f} TB/s read")
