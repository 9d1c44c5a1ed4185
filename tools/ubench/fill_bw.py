import torch, time
x=torch.empty(int(5.6e9)//4, device='cuda')
for f in (lambda: x.fill_(1.0), lambda: x.zero_()):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(10): f()
    torch.cuda.synchronize(); dt=(time.time()-t)/10
    print('fill %.3f ms  %.2f TB/s'%(dt*1e3, x.numel()*4/dt/1e12))
y=torch.empty_like(x[:x.numel()//2]); 
for _ in range(3): y.copy_(x[:y.numel()])
torch.cuda.synchronize(); t=time.time()
for _ in range(10): y.copy_(x[:y.numel()])
torch.cuda.synchronize(); dt=(time.time()-t)/10
print('copy %.3f ms  %.2f TB/s (read+write)'%(dt*1e3, 2*y.numel()*4/dt/1e12))
