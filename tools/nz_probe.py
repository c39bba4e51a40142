import torch, time
dev="cuda:0"
x=(torch.rand(12*218*1153,device=dev)<0.06)
v=torch.rand(x.numel(),device=dev)
cap=x.numel()//6
def t(name,fn,n=20):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print("%-28s host %.3f ms/call, total %.3f ms/call"%(name,(t1-t0)/n*1e3,(t2-t0)/n*1e3))
t("nonzero_static", lambda: torch.nonzero_static(x,size=cap,fill_value=x.numel()))
def comp():
    pos=torch.cumsum(x.to(torch.int32),0)-1
    dest=torch.where(x&(pos<cap),pos,cap).long()
    return torch.zeros(cap+1,device=dev).scatter_(0,dest,v)[:cap]
t("cumsum+scatter", comp)
t("sort cap", lambda: torch.sort(v[:cap]))
t("sort 3M", lambda: torch.sort(v))
t("mask sum", lambda: x.sum())
