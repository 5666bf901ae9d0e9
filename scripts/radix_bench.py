"""Micro-benchmark of one radix pass: python scripts/radix_bench.py [count] [algo]"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from debwt_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
algo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator(device="cuda").manual_seed(1)
keys = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=g)
work = keys.clone(); tmp = torch.empty_like(keys)
d = api.DeBWT(k=32, sort_algo=algo)
for it in range(3):
    work.copy_(keys); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms = d.radix_sort_device(work.data_ptr(), tmp.data_ptr(), n, 64, want_ms=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print(f"algo {algo} n={n}: total {dt:.2f} ms, scatter pass {ms:.3f} ms = {16*n/ms/1e6:.0f} GB/s algorithmic", flush=True)
u = work ^ (-2**63); assert bool((u[1:] >= u[:-1]).all())
