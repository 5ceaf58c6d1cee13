import torch, sys
sys.path.insert(0, "/root/repo")
dev = torch.device("cuda", 0)
for n in (838_860_800, 209_715_200):
    y = torch.empty(n // 2, dtype=torch.bfloat16, device=dev).normal_()
    z = torch.empty_like(y)
    for _ in range(3): z.copy_(y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): z.copy_(y)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"copy {n/1e6:.0f} MB: {t*1e3:.1f} us  {2*n/t/1e6:.0f} GB/s")
    e0.record()
    for _ in range(10): z.mul_(1.0001)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"inplace mul {n/1e6:.0f} MB: {t*1e3:.1f} us  {2*n/t/1e6:.0f} GB/s")
    e0.record()
    for _ in range(10): torch.add(y, 1.0, out=z)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"add out {n/1e6:.0f} MB: {t*1e3:.1f} us  {2*n/t/1e6:.0f} GB/s")
