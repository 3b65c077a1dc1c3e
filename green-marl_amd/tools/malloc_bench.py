import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
def t(nbytes, n=3):
    out = []
    for _ in range(n):
        p = ctypes.c_void_p()
        t0 = time.perf_counter(); r = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)); t1 = time.perf_counter()
        hip.hipMemset(p, 0, ctypes.c_size_t(nbytes)); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
        hip.hipFree(p); t3 = time.perf_counter()
        out.append((t1-t0, t2-t1, t3-t2))
    return out
hip.hipSetDevice(0)
for gb in (0.001, 0.25, 1, 4, 8, 16):
    print(gb, ["malloc %.2f ms memset %.2f ms free %.2f ms" % tuple(x*1e3 for x in r) for r in t(int(gb*(1<<30)))])
