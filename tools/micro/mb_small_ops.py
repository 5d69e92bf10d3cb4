import torch, time
dev=torch.device("cuda:0")
x=torch.rand(100,100,device=dev)
def t(f,n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
print("topk ms", t(lambda: torch.topk(x,5,dim=-1,largest=False)))
print("sort ms", t(lambda: torch.sort(x,dim=-1)))
print("argsort[:5] ms", t(lambda: torch.argsort(x,dim=-1)[:,:5]))
c=torch.rand(100,3)
print("cpu max ms", t(lambda: c.max(dim=1,keepdim=True)))
print("cpu det ms", t(lambda: torch.linalg.det(torch.rand(100,3,3))))
print("threads", torch.get_num_threads())
import numpy as np
cn=c.numpy()
print("np max ms", t(lambda: cn.max(1)))
