import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dev=torch.device("cuda",0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x=torch.arange(1<<20, device=dev, dtype=torch.float32)
dist.all_reduce(x); torch.cuda.synchronize()
print("nccl world1 all_reduce ok", x[:3].tolist(), dist.get_backend())
# bucketed path with a side stream
import sys; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
from refil_amd import dp
g=torch.ones(1000, device=dev)
dp.allreduce_sum_(g); torch.cuda.synchronize(); print("dp.allreduce_sum_ ok", g[:2].tolist(), dp.world())
dist.destroy_process_group()
