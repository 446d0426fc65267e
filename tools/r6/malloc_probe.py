"""How long do hipMalloc / hipFree of the sizes a first minibatch asks for take?  (round 6: the
host side of the first-touch build)"""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipDeviceSynchronize()
p = C.c_void_p()
hip.hipMalloc(C.byref(p), 1 << 20)
hip.hipFree(p)
for mb in (1.6, 25, 50, 90, 120, 400):
    n = int(mb * (1 << 20))
    ta, tf = [], []
    for _ in range(6):
        t0 = time.perf_counter()
        hip.hipMalloc(C.byref(p), n)
        t1 = time.perf_counter()
        hip.hipFree(p)
        t2 = time.perf_counter()
        ta.append((t1 - t0) * 1e6)
        tf.append((t2 - t1) * 1e6)
    print("%6.1f MB  hipMalloc %s us   hipFree %s us" % (mb, [round(x) for x in ta], [round(x) for x in tf]))
