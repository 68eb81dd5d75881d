import torch, time
dev="cuda"
M,K,N=32768,640,2560
x=torch.randn(M,K,device=dev,dtype=torch.float16)
w=torch.randn(N,K,device=dev,dtype=torch.float16)*0.04
f8=torch.float8_e4m3fn
sx=(x.abs().amax(dim=1,keepdim=True).float()/448.0)
sw=(w.abs().amax(dim=1,keepdim=True).float()/448.0)
xq=(x.float()/sx).to(f8); wq=(w.float()/sw).to(f8)
ref=(x.float()@w.float().t())
for name,kw in [("rowwise",dict(scale_a=sx,scale_b=sw.t().contiguous())),("tensorwise",dict(scale_a=sx.max().reshape(()) ,scale_b=sw.max().reshape(())))]:
    try:
        if name=="tensorwise":
            xq2=(x.float()/kw["scale_a"]).to(f8); wq2=(w.float()/kw["scale_b"]).to(f8)
            out=torch._scaled_mm(xq2,wq2.t(),out_dtype=torch.float16,**kw)
        else:
            out=torch._scaled_mm(xq,wq.t(),out_dtype=torch.float16,**kw)
        err=(out.float()-ref).abs().max().item()/ref.abs().max().item()
        torch.cuda.synchronize(); t=time.time()
        for _ in range(50):
            out=torch._scaled_mm(xq2 if name=="tensorwise" else xq,(wq2 if name=="tensorwise" else wq).t(),out_dtype=torch.float16,**kw)
        torch.cuda.synchronize(); dt=(time.time()-t)/50*1e6
        print(name,"ok rel err",err,"us",dt, "TF/s", 2*M*K*N/dt/1e6)
    except Exception as e:
        print(name,"FAILED",repr(e)[:300])
torch.cuda.synchronize(); t=time.time()
for _ in range(50): o=x@w.t()
torch.cuda.synchronize(); dt=(time.time()-t)/50*1e6
print("fp16 us",dt,"TF/s",2*M*K*N/dt/1e6)
