import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
if os.environ.get('NCCL_DEBUG', '').upper() == 'VERSION':
  os.environ['NCCL_DEBUG'] = 'NONE'   # as bench.py: no RCCL banner on stdout
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from dqn_zoo_amd import distributed as dz
st = dz.ReplicaStats(dev)
st.add(grad_steps=7, loss_sum=torch.tensor(3.5, dtype=torch.float64, device=dev))
tot = st.all_reduce()
torch.cuda.synchronize()
print('RCCL world=1 all_reduce ok', {k: float(v) for k, v in tot.items() if k in ('grad_steps', 'loss_sum', 'replicas')})
dist.destroy_process_group()
