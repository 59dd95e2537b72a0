import os, torch, torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
t = torch.ones(4, device=dev, dtype=torch.float64) * (dist.get_rank() + 1)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
print("nccl ok", t.tolist(), dist.get_world_size())
dist.destroy_process_group()
