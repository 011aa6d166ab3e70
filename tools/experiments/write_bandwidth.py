"""What can a write-only stream reach on this GPU?  (k_bsr_fill writes 1.88 GB and reads 0.56 GB per launch.)
torch fill_ / copy_ of 2 GB buffers, HIP-event timed."""
import torch

dev = torch.device("cuda:0")
n = 2 * 1024 ** 3 // 8
a = torch.empty(n, dtype=torch.float64, device=dev)
b = torch.empty(n, dtype=torch.float64, device=dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timed(lambda: a.fill_(1.5))
print(f"fill_  2 GiB: {ms:.3f} ms = {2 * 1024 ** 3 / ms / 1e9:.2f} TB/s written")
ms = timed(lambda: b.copy_(a))
print(f"copy_  2 GiB: {ms:.3f} ms = {2 * 1024 ** 3 / ms / 1e9:.2f} TB/s written + the same read")
ms = timed(lambda: torch.add(a, 1.0, out=b))
print(f"add    2 GiB: {ms:.3f} ms = {2 * 1024 ** 3 / ms / 1e9:.2f} TB/s written + the same read")
