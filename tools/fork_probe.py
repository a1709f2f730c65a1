import sys, time, torch
sys.path.insert(0,'.')
from tools.kbench import mk, actions
for n in (1<<12, 1<<16, 1<<20):
    e = mk("c2", n, spec=False); a = actions(e, n)
    for _ in range(5): e.step(a)
    torch.cuda.synchronize()
    t0=time.perf_counter(); cs=[e.fork() for _ in range(5)]; torch.cuda.synchronize(); t_new=(time.perf_counter()-t0)/5
    c=cs[0]
    t0=time.perf_counter()
    for _ in range(50): e.fork(into=c)
    torch.cuda.synchronize(); t_into=(time.perf_counter()-t0)/50
    print(f"N={n}: fork() new {t_new*1e3:.2f} ms | fork(into=) {t_into*1e6:.1f} us")
