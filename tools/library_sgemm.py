import torch, time
for n in (4096, 8192):
    a=torch.randn(n,n,device="cuda"); b=torch.randn(n,n,device="cuda")
    for _ in range(3): c=a@b
    torch.cuda.synchronize(); t0=time.perf_counter()
    it=20
    for _ in range(it): c=a@b
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/it
    print(f"torch.matmul fp32 {n}^3: {2*n**3/dt/1e12:.1f} TFLOP/s")
# conv-shaped: (2400 x 3072) @ (3072 x 1024)
a=torch.randn(4800,3072,device="cuda"); b=torch.randn(3072,1024,device="cuda")
for _ in range(3): c=a@b
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(50): c=a@b
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/50
print(f"torch.matmul fp32 4800x3072x1024: {2*4800*3072*1024/dt/1e12:.1f} TFLOP/s ({dt*1e6:.0f} us)")
