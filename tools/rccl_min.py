import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29777"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
print("init ok", flush=True)
dist.barrier(); torch.cuda.synchronize(); print("barrier ok", flush=True)
t=torch.tensor([1.5],dtype=torch.float64,device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); torch.cuda.synchronize(); print("allreduce ok", float(t.item()), flush=True)
dist.destroy_process_group(); print("done")
