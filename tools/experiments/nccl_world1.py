"""The three torch.distributed calls bench.py makes at N > 1 (all_reduce MAX of a float64, all_gather_into_tensor of
int32 rows, barrier) on the nccl (= RCCL) backend with a world of one rank: checks the API use on this image."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t=torch.tensor([1.5],dtype=torch.float64,device=dev); dist.all_reduce(t,op=dist.ReduceOp.MAX)
p=torch.arange(20,dtype=torch.int32,device=dev).view(5,4); g=torch.empty((5,4),dtype=torch.int32,device=dev)
dist.all_gather_into_tensor(g,p); dist.barrier(); torch.cuda.synchronize()
print("nccl world-1 ok", float(t.item()), bool(torch.equal(g,p)))
dist.destroy_process_group()
