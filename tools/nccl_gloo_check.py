"""diagnostic: the process-group combination of the pipelined sharded pass — nccl default group + gloo side group on the
loopback interface — comes up on this box and both carry a collective (world size 1 is all a one-GPU box allows)"""
import os, sys
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
g = dist.new_group(backend="gloo")
a = torch.tensor([7], dtype=torch.int64); out = torch.empty(1, dtype=torch.int64)
dist.all_gather_into_tensor(out, a, group=g)
r = torch.empty(1, dtype=torch.int64); dist.all_to_all_single(r, a, group=g)
x = torch.arange(10, dtype=torch.int64, device=dev); y = torch.empty(10, dtype=torch.int64, device=dev)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    z = x * 2
torch.cuda.current_stream().wait_stream(s)
dist.all_to_all([y], [z])
torch.cuda.synchronize()
print("gloo side group:", out.tolist(), r.tolist(), " nccl all_to_all:", y[:3].tolist(), "ok")
dist.destroy_process_group()
