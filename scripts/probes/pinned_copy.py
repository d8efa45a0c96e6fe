"""How fast does a history-sized device-to-host copy go: pageable numpy memory, the same registered (pinned) first, and what
does registering cost?  (plain HIP runtime calls through ctypes; the library is not involved)"""
import ctypes as C, time
import numpy as np
hip = C.CDLL("libamdhip64.so")
n = 492 * 1000 * 1000
dev = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), C.c_size_t(n)) == 0
hip.hipMemset(dev, 1, C.c_size_t(n)); hip.hipDeviceSynchronize()
for rep in range(3):
    a = np.empty(n, dtype=np.uint8)
    t0 = time.perf_counter()
    assert hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), dev, C.c_size_t(n), 2) == 0
    t1 = time.perf_counter()
    print(f"pageable, fresh array: {1e3*(t1-t0):.1f} ms = {n/1e9/(t1-t0):.1f} GB/s")
    t0 = time.perf_counter()
    assert hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), dev, C.c_size_t(n), 2) == 0
    t1 = time.perf_counter()
    print(f"pageable, touched array: {1e3*(t1-t0):.1f} ms = {n/1e9/(t1-t0):.1f} GB/s")
    b = np.empty(n, dtype=np.uint8)
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(b.ctypes.data_as(C.c_void_p), C.c_size_t(n), 0)
    t1 = time.perf_counter()
    assert hip.hipMemcpy(b.ctypes.data_as(C.c_void_p), dev, C.c_size_t(n), 2) == 0
    t2 = time.perf_counter()
    hip.hipHostUnregister(b.ctypes.data_as(C.c_void_p))
    t3 = time.perf_counter()
    print(f"register rc={rc}: {1e3*(t1-t0):.1f} ms, copy {1e3*(t2-t1):.1f} ms = {n/1e9/(t2-t1):.1f} GB/s, unregister {1e3*(t3-t2):.1f} ms")
