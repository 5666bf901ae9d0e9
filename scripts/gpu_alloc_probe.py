"""What a process pays for device memory right after another process released it (the cold path of the one-shot program).
python scripts/gpu_alloc_probe.py            parent: runs child A (allocates + touches GB, exits), then child B (times hipMalloc)"""
import ctypes, os, subprocess, sys, time
hip = ctypes.CDLL("libamdhip64.so")
def malloc(nbytes):
    p = ctypes.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)); dt = time.perf_counter() - t0
    return rc, p, dt
def meminfo():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 1e9, t.value / 1e9
if len(sys.argv) > 1 and sys.argv[1] == "A":
    gb = int(sys.argv[2]); ps = []
    for i in range(gb // 25):
        rc, p, dt = malloc(25 << 30); ps.append(p)
        hip.hipMemset(p, 1, ctypes.c_size_t(25 << 30))
    hip.hipDeviceSynchronize()
    print(f"A: holds {gb // 25 * 25} GiB, free now {meminfo()[0]:.1f} GB", flush=True)
elif len(sys.argv) > 1 and sys.argv[1] == "B":
    t0 = time.perf_counter()
    hip.hipInit(0)
    print(f"B: start, free {meminfo()[0]:.1f} of {meminfo()[1]:.1f} GB (init {time.perf_counter() - t0:.2f} s)", flush=True)
    tot = 0.0
    for i in range(int(sys.argv[2]) // 25):
        rc, p, dt = malloc(25 << 30); tot += dt
        print(f"B: hipMalloc #{i} of 25 GiB: rc {rc}, {dt * 1e3:.1f} ms, free {meminfo()[0]:.1f} GB, t = {time.perf_counter() - t0:.2f} s", flush=True)
    t1 = time.perf_counter()
    hip.hipMemset(p, 0, ctypes.c_size_t(25 << 30)); hip.hipDeviceSynchronize()
    print(f"B: total malloc {tot:.2f} s; first touch of the last 25 GiB {time.perf_counter() - t1:.3f} s", flush=True)
else:
    me = os.path.abspath(__file__)
    for gbA, gbB, pause in ((250, 250, 0.0), (250, 100, 0.0), (250, 250, 5.0), (0, 250, 0.0)):
        if gbA: subprocess.run([sys.executable, me, "A", str(gbA)])
        time.sleep(pause)
        print(f"--- after a process that held {gbA} GiB, pause {pause} s: allocate {gbB} GiB", flush=True)
        subprocess.run([sys.executable, me, "B", str(gbB)])
