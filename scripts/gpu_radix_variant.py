"""One radix pass on random keys for each variant lib: python scripts/gpu_radix_variant.py NAME...  (D* = diagnostic
variants with wrong results on purpose, not checked)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(name, n):
    import torch
    from debwt_amd import api
    g = torch.Generator(device="cuda").manual_seed(1)
    keys = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=g)
    work = keys.clone(); tmp = torch.empty_like(keys)
    d = api.DeBWT(k=32, sort_algo=1)
    res = []
    for it in range(3):
        work.copy_(keys); torch.cuda.synchronize()
        res.append(d.radix_sort_device(work.data_ptr(), tmp.data_ptr(), n, 64, want_ms=True))
    ok = ""
    if not name.startswith("D"):
        u = work ^ (-2**63); ok = " sorted=%s" % bool((u[1:] >= u[:-1]).all())
    print(f"{name}: scatter pass {min(res):.3f} ms = {16*n/min(res)/1e6:.0f} GB/s{ok}", flush=True)

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        for name in sys.argv[1:]:
            env = dict(os.environ)
            if name != "default":
                env["DEBWT_HIP_LIB"] = os.path.join(ROOT, "build", "variants", f"libdebwt_{name}.so")
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", name, os.environ.get("NKEYS", "250000000")], env=env)
