"""cost of hipHostRegister / hipHostUnregister on pageable buffers of a few sizes (would pinning the caller's sequences beat gathering them?)"""
import ctypes as C, time, numpy as np
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]; hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipSetDevice(0); p = C.c_void_p(); hip.hipMalloc(C.byref(p), 1 << 28)
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
for mb in (0.01, 0.1, 1, 5, 50):
    n = int(mb * (1 << 20)); bufs = [np.random.randint(0, 255, n, dtype=np.uint8) for _ in range(8)]
    t0 = time.perf_counter()
    for b in bufs: assert hip.hipHostRegister(b.ctypes.data, n, 0) == 0
    t1 = time.perf_counter()
    for b in bufs: assert hip.hipMemcpy(p, b.ctypes.data, n, 1) == 0
    t2 = time.perf_counter()
    for b in bufs: assert hip.hipHostUnregister(b.ctypes.data) == 0
    t3 = time.perf_counter()
    bufs2 = [np.random.randint(0, 255, n, dtype=np.uint8) for _ in range(8)]
    t4 = time.perf_counter()
    for b in bufs2: assert hip.hipMemcpy(p, b.ctypes.data, n, 1) == 0
    t5 = time.perf_counter()
    print(f"{mb:6.2f} MB: register {1e6*(t1-t0)/8:8.1f} us  copy from registered {1e6*(t2-t1)/8:8.1f} us  unregister {1e6*(t3-t2)/8:8.1f} us | copy from pageable {1e6*(t5-t4)/8:8.1f} us")
