"""RCCL sanity on the GPU box: single-rank process group, all_reduce, all_gather, barrier (the calls bench.py makes for N > 1)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29577")
os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda",0))
t=torch.tensor([1.5],dtype=torch.float64,device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX); g=[torch.zeros_like(t)]; dist.all_gather(g,t); dist.barrier()
print("nccl ok", t.item(), g[0].item()); dist.destroy_process_group()
