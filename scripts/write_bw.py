"""Reference point for write-stream kernels: what a plain fill of a 2 GB tensor reaches on this GPU (torch's fill kernel)."""
import torch
x = torch.empty(131072 * 32 * 121, dtype=torch.float32, device="cuda")
for _ in range(3):
    x.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    x.zero_()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print({"bytes": x.numel() * 4, "ms": ms, "GBps": x.numel() * 4 / (ms * 1e-3) / 1e9})
