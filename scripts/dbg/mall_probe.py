"""Does a read that follows a write of the same buffer come out of the 256-MB Infinity Cache?  (the question behind sub-batch
interleaving of convolution and BatchNorm statistics, VERDICT r2 task 5a)  torch ops only: fill (write N bytes), then a reduction
(read N bytes) timed with events, hot (right behind the write) and cold (after 2 GB of other traffic)."""
import torch
dev = torch.device("cuda", 0)
scrub = torch.empty(2 << 30, dtype=torch.uint8, device=dev)


def t_read(n_mb, hot):
    x = torch.empty(n_mb << 20, dtype=torch.uint8, device=dev).view(torch.float32)
    best = 1e9
    for _ in range(5):
        x.fill_(1.0)
        if not hot:
            scrub.fill_(3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s = x.sum()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for n in (16, 32, 64, 128, 192, 256, 384, 512, 1024):
    h, c = t_read(n, True), t_read(n, False)
    print(f"{n:5d} MB: read behind its own write {n / 1024 / h * 1e3:6.2f} TB/s ({h * 1e3:7.1f} us)   cold {n / 1024 / c * 1e3:6.2f} TB/s ({c * 1e3:7.1f} us)")
