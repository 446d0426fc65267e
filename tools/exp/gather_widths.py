"""N = 1e7 random gathers of W bytes from a table of S bytes (uniform indices): microseconds.
Design input for the FM forward."""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "gather_widths.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                       "-o", so, os.path.join(HERE, "gather_widths.hip")])
X = C.CDLL(so)
torch.cuda.set_device(0)
N = 10_000_000
s = torch.cuda.current_stream().cuda_stream
out = torch.empty(8192 * 256, device="cuda")
vp = C.c_void_p
print("%8s %10s | %s" % ("W bytes", "table", "us per 1e7 gathers (grid 2048 / 8192)"))
for W in (4, 8, 16, 32, 64):
    for rows in (50_000, 1_000_000, 6_320_000, 10_000_000):
        S = rows * W
        if S > (1 << 31):
            continue
        tab = torch.zeros(S // 4, dtype=torch.float32, device="cuda")
        idx = torch.randint(0, rows, (N,), dtype=torch.int32, device="cuda")
        res = []
        for grid in (2048, 8192):
            for _ in range(3):
                X.x_gather(W, vp(idx.data_ptr()), vp(tab.data_ptr()), C.c_size_t(N), vp(out.data_ptr()), grid, vp(s))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                X.x_gather(W, vp(idx.data_ptr()), vp(tab.data_ptr()), C.c_size_t(N), vp(out.data_ptr()), grid, vp(s))
            e1.record(); e1.synchronize()
            res.append(e0.elapsed_time(e1) / 10 * 1e3)
        print("%8d %8.1fMB | %7.1f %7.1f" % (W, S / 1e6, res[0], res[1]), flush=True)
        del tab, idx
