import torch, json
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (42, 69, 111, 512, 2048, 8192):
    nel = mb * 1024 * 1024 // 4
    a = torch.empty(nel, dtype=torch.float32, device=dev); b = torch.ones(nel, dtype=torch.float32, device=dev)
    tw = timeit(lambda: a.fill_(1.0)); tc = timeit(lambda: a.copy_(b)); tr = timeit(lambda: b.sum())
    print(f"{mb:5d} MB  write-only {mb/1024/tw:7.2f} GiB/s-> {mb*1.048576e6/tw/1e12:5.2f} TB/s | copy (R+W) {2*mb*1.048576e6/tc/1e12:5.2f} TB/s | read-only(sum) {mb*1.048576e6/tr/1e12:5.2f} TB/s")
    del a, b
