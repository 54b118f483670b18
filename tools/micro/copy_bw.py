"""HBM streaming reference points (torch elementwise kernels): what a pure copy / read-only pass reaches on this GPU.
Used to judge the 1x1-conv layers, which move K+N bytes per pixel and do almost no math."""
import torch, sys
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for mb in (26, 105, 420, 1680):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device=dev, dtype=torch.float16); y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print(f"copy   {mb:5d} MB read + {mb:5d} MB write: {t*1e6:8.1f} us  {2*mb*1.048576e6/t/1e12:6.2f} TB/s", flush=True)
    t = timeit(lambda: torch.relu_(y))
    print(f"relu_  {mb:5d} MB read + {mb:5d} MB write (in place): {t*1e6:8.1f} us  {2*mb*1.048576e6/t/1e12:6.2f} TB/s", flush=True)
    t = timeit(lambda: x.sum())
    print(f"sum    {mb:5d} MB read: {t*1e6:8.1f} us  {mb*1.048576e6/t/1e12:6.2f} TB/s", flush=True)
    t = timeit(lambda: y.zero_())
    print(f"zero   {mb:5d} MB write: {t*1e6:8.1f} us  {mb*1.048576e6/t/1e12:6.2f} TB/s", flush=True)
    del x, y
