"""Does the first large kernel after a short idle period run slower?  (bench.py's timed pass starts after a barrier + sync.)
Times a 12 GB fill after idle periods of different lengths, with events on the stream."""
import time
import torch

x = torch.empty(3 * 1024 ** 3, dtype=torch.float32, device="cuda")
x.fill_(0.0)
torch.cuda.synchronize()
for idle_ms in (0, 0, 1, 2, 5, 10, 20, 50, 100, 300, 1000, 0, 0):
    torch.cuda.synchronize()
    time.sleep(idle_ms / 1e3)
    a, b, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    x.fill_(1.0)
    b.record()
    x.fill_(2.0)
    c.record()
    torch.cuda.synchronize()
    print(f"idle {idle_ms:5d} ms: first fill {a.elapsed_time(b):7.2f} ms, second {b.elapsed_time(c):7.2f} ms")
