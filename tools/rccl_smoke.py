import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29511")
os.environ["RANK"]="0"; os.environ["WORLD_SIZE"]="1"; os.environ["LOCAL_RANK"]="0"
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
x=torch.arange(6,dtype=torch.float32,device="cuda").view(2,3)
out=torch.empty((2,3),device="cuda"); dist.all_gather_into_tensor(out,x)
n=torch.tensor([2],dtype=torch.int64,device="cuda"); l=[torch.zeros_like(n)]; dist.all_gather(l,n)
t=torch.tensor([1.5],dtype=torch.float64,device="cuda"); dist.all_reduce(t,op=dist.ReduceOp.MAX); dist.barrier()
print("rccl single-rank ok", out.tolist(), l[0].item(), t.item())
dist.destroy_process_group()
