"""RCCL behaviour of the collectives sharded.py uses, in a process group of one (the only group a 1-GPU box allows):
does all_to_all_single / all_to_all / all_gather_into_tensor / gather move large buffers completely?"""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
def check(name, fn, n):
    src = torch.arange(n, dtype=torch.int64, device="cuda") * 3 + 1
    dst = torch.zeros(n + 64, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn(dst, src, n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    bad = (dst[:n] != src).nonzero().flatten()
    print(f"n={n} {name}: wrong {bad.numel()} first {bad[:1].tolist()} ({dt*1e3:.1f} ms)", flush=True)
def a2a_single_splits(dst, src, n): dist.all_to_all_single(dst[:n], src[:n], output_split_sizes=[n], input_split_sizes=[n])
def a2a_single_equal(dst, src, n): dist.all_to_all_single(dst[:n], src[:n])
def a2a_list(dst, src, n): dist.all_to_all([dst[:n]], [src[:n]])
def p2p(dst, src, n):
    ops = [dist.P2POp(dist.isend, src[:n], 0), dist.P2POp(dist.irecv, dst[:n], 0)]
    for w in dist.batch_isend_irecv(ops): w.wait()
def gather(dst, src, n): dist.gather(src[:n], [dst[:n]], dst=0)
def a2a_chunked(dst, src, n, ch=1 << 26):
    for o in range(0, n, ch):
        m = min(ch, n - o)
        dist.all_to_all_single(dst[o:o + m], src[o:o + m], output_split_sizes=[m], input_split_sizes=[m])
for n in (130_000_000, 134_217_727, 134_217_728, 140_000_000, 300_000_000):
    for name, fn in (("all_to_all_single(splits)", a2a_single_splits), ("all_to_all_single(equal)", a2a_single_equal),
                     ("all_to_all(list)", a2a_list), ("batch_isend_irecv", p2p), ("gather", gather), ("chunked 2^26", a2a_chunked)):
        try:
            check(name, fn, n)
        except Exception as e:
            print(f"n={n} {name}: {type(e).__name__} {str(e)[:120]}", flush=True)
dist.destroy_process_group()
