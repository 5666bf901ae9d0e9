import time, torch
torch.cuda.init()
for gb in (1, 8, 32, 32, 32):
    torch.cuda.synchronize(); t0=time.time()
    x=torch.empty(int(gb*(1<<30)), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1=time.time()
    x.zero_(); torch.cuda.synchronize(); t2=time.time()
    x.zero_(); torch.cuda.synchronize(); t3=time.time()
    print(f"{gb} GiB: alloc {1e3*(t1-t0):.1f} ms, first touch {1e3*(t2-t1):.1f} ms, second {1e3*(t3-t2):.1f} ms", flush=True)
    if gb < 32: del x; torch.cuda.empty_cache()
