import torch,time
x=torch.empty(1<<28,dtype=torch.float32,device='cuda')  # 1 GiB
for _ in range(3): x.fill_(1.0)
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): x.fill_(1.0)
e1.record(); torch.cuda.synchronize()
print("fill 1 GiB: %.0f GB/s"%(x.numel()*4*10/e0.elapsed_time(e1)/1e6))
y=torch.empty_like(x)
e0.record()
for _ in range(10): y.copy_(x)
e1.record(); torch.cuda.synchronize()
print("copy 1 GiB: %.0f GB/s r+w"%(2*x.numel()*4*10/e0.elapsed_time(e1)/1e6))
