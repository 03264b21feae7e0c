"""Control experiment for the two-ranks-on-one-GPU rehearsal of bench.py over gloo: the reducer's collective pattern alone
(13 asynchronous all-reduces of 9 MB slices of one flat buffer per step, waited for at the end of the step), no model."""
import faulthandler
import os

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(60, exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 29_200_000
g = torch.zeros(n, device="cuda")
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda")
for step in range(4):
    g.fill_(float(rank + 1))
    pend = []
    per = n // 13
    for i in range(13):
        b = a @ a                                  # compute between the bucket reports
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pend.append(dist.all_reduce(g[i * per:(i + 1) * per], async_op=True))
    for h in pend:
        h.wait()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print(rank, "step", step, "ok", float(g[0]), float(g[per * 12 + 5]), flush=True)
dist.barrier()
dist.destroy_process_group()
