import torch, time
dev='cuda'
for mb in (8, 16, 32, 64, 128, 256, 512, 2048):
    n = mb*1024*1024//4
    x = torch.empty(n, device=dev).normal_(); y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    reps = max(5, 4096//mb)
    e0.record()
    for _ in range(reps): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    print(f"copy {mb:5d} MB -> {mb} MB: {ms*1e3:8.1f} us  {2*mb/1024/ms*1e3/1000:6.2f} TB/s (read+write)")
    # write-only
    e0.record()
    for _ in range(reps): y.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    print(f"fill {mb:5d} MB: {ms*1e3:8.1f} us  {mb/1024/ms*1e3/1000:6.2f} TB/s")
