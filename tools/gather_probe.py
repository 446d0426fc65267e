import torch, numpy as np
torch.cuda.set_device(0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/n*1e3
N=10_000_000
for tab in (50_000, 790_000, 6_320_000):
    table=torch.randn(tab,device="cuda")
    idx=torch.randint(0,tab,(N,),device="cuda",dtype=torch.int64)
    idx32=idx.to(torch.int32)
    out=torch.empty(N,device="cuda")
    print("table %8d floats: index_select i64 %.1f us ; take i32->gather %.1f us" % (tab, t(lambda: torch.index_select(table,0,idx,out=out)), t(lambda: torch.index_select(table,0,idx32,out=out))))
    sidx=torch.sort(idx32).values
    print("      sorted idx: %.1f us" % t(lambda: torch.index_select(table,0,sidx,out=out)))
a=torch.empty(N,device="cuda"); b=torch.empty(N,device="cuda")
print("copy 40MB: %.1f us" % t(lambda: b.copy_(a)))
