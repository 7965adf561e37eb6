import torch, time
dev="cuda:0"
for (M,K,N) in [(8192,4096,4096),(8192,4096,27392),(8192,13696,4096),(2048,4096,4096),(512,4096,4096)]:
    a=torch.randn(M,K,device=dev,dtype=torch.float16)
    ws=[torch.randn(K,N,device=dev,dtype=torch.float16) for _ in range(4)]
    wts=[w.t().contiguous() for w in ws]
    for name,fn in (("A@W(KxN)",lambda i: a@ws[i]),("A@Wt.T(NxK)",lambda i: a@wts[i].t())):
        for i in range(4): fn(i)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(5):
            for i in range(4): fn(i)
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/20
        print(M,K,N,name, round(ms*1e3,1),"us", round(2*M*N*K/ms/1e9,1),"TF")
