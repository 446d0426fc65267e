import ctypes as C, os, subprocess, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
so = os.path.join(HERE, "exp_kernels.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "exp_kernels.hip")])
X = C.CDLL(so)
from xflow_amd import capi
torch.cuda.set_device(0)
R, nnz, K = 50000, 200, 10_000_000
rng = np.random.RandomState(1)
keytab = capi.hash_decimal_range(0, K)
fid = rng.randint(0, K, size=R * nnz)
b = capi.Batch(np.arange(R + 1, dtype=np.uint64) * np.uint64(nnz), keytab[fid], rng.randint(0, 2, size=R).astype(np.int32))
h = b.host(); tp = b.tiles()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()
segptr, coo, tilep = dev(h["segptr"]), dev(h["coo_row"]), dev(tp)
# wave tiles: <=512 nnz, <= 512 keys
wt = [0]; st = 0; n = 0
seg = np.diff(h["segptr"])
for u in range(b.U):
    if n + seg[u] > 512 or u - st == 512:
        wt.append(u); st = u; n = 0
    n += seg[u]
wt.append(b.U); wtp = dev(np.array(wt, dtype=np.uint32))
loss = torch.randn(R, device="cuda"); out = torch.empty(b.NNZ, device="cuda"); g = torch.empty(b.U, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def t(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    print("%-40s %7.1f us" % (name, e0.elapsed_time(e1) / n * 1e3), flush=True)
vp = C.c_void_p
for grid in (2048, 8192):
    t("flat gather+store grid %d" % grid, lambda: X.x_flat(vp(coo.data_ptr()), vp(loss.data_ptr()), C.c_size_t(b.NNZ), vp(out.data_ptr()), grid, vp(s)))
    t("flat gather nostore grid %d" % grid, lambda: X.x_flat_nostore(vp(coo.data_ptr()), vp(loss.data_ptr()), C.c_size_t(b.NNZ), vp(out.data_ptr()), grid, vp(s)))
nt = len(tp) - 1
for grid in (1024, 2560, nt):
    t("stage256 grid %d" % grid, lambda: X.x_stage256(vp(tilep.data_ptr()), nt, vp(segptr.data_ptr()), vp(coo.data_ptr()), vp(loss.data_ptr()), vp(out.data_ptr()), grid, vp(s)))
    t("tiled256 grid %d" % grid, lambda: X.x_tiled256(vp(tilep.data_ptr()), nt, vp(segptr.data_ptr()), vp(coo.data_ptr()), vp(loss.data_ptr()), vp(g.data_ptr()), grid, vp(s)))
for grid in (256, 512):
    t("tiled1024 grid %d" % grid, lambda: X.x_tiled1024(vp(tilep.data_ptr()), nt, vp(segptr.data_ptr()), vp(coo.data_ptr()), vp(loss.data_ptr()), vp(g.data_ptr()), grid, vp(s)))
nwt = len(wt) - 1
for grid in (2048, 4096):
    t("wavetile grid %d (nwt %d)" % (grid, nwt), lambda: X.x_wavetile(vp(wtp.data_ptr()), nwt, vp(segptr.data_ptr()), vp(coo.data_ptr()), vp(loss.data_ptr()), vp(g.data_ptr()), grid, vp(s)))

for grid in (256, 512):
    t("wavetile_lds grid %d" % grid, lambda: X.x_wavetile_lds(vp(wtp.data_ptr()), nwt, vp(segptr.data_ptr()), vp(coo.data_ptr()), vp(loss.data_ptr()), R, vp(g.data_ptr()), grid, vp(s)))
gref = torch.empty_like(g)
X.x_tiled256(vp(tilep.data_ptr()), nt, vp(segptr.data_ptr()), vp(coo.data_ptr()), vp(loss.data_ptr()), vp(gref.data_ptr()), nt, vp(s))
torch.cuda.synchronize()
print("wavetile_lds == tiled256:", bool(torch.equal(g, gref)))
