"""Achievable HBM write / copy bandwidth with PyTorch's own kernels (fill, copy) on 2 GiB buffers: the yardstick for the
training kernels' 3.2-3.3 TB/s of kept-activation stores (profiles/r04_ab_notes.txt)."""
import torch, time
n = 512 * 1024 * 1024
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
for name, f, nbytes in (("zero_", lambda: x.zero_(), 4 * n), ("fill_", lambda: x.fill_(1.5), 4 * n),
                        ("copy_ (read + write)", lambda: y.copy_(x), 8 * n), ("sum (read)", lambda: x.sum(), 4 * n),
                        ("mul_ (read + write in place)", lambda: x.mul_(1.0001), 8 * n)):
    t = timed(f)
    print(f"{name}: {nbytes / t / 1e12:.2f} TB/s ({t * 1e3:.3f} ms)", flush=True)
