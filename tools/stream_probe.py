"""Which hardware queue do torch's pool streams share?  Prints the sharing matrix measured by ops.streams_share_queue
(main + 10 pool streams), plain and after an RCCL process group exists.  usage: python tools/stream_probe.py [nccl]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from geomae_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if "nccl" in sys.argv:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    t = torch.zeros(4, device=dev)
    g2 = dist.new_group(pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
    dist.all_reduce(t)
    dist.all_reduce(t, group=g2, async_op=True).wait()
    torch.cuda.synchronize()
torch.zeros(1, device=dev)
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(10)] + \
    [torch.cuda.Stream(device=dev, priority=-1) for _ in range(3)]
for s in streams:
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
torch.cuda.synchronize()
cyc = ops._spin_us(dev)
n = len(streams)
alone = min(ops._spin_seconds(s, cyc, dev) for s in streams)
print("spin cycles", cyc, f"alone {alone * 1e6:.0f} us; rows: WAITING stream, columns: VICTIM stream; helper = a third stream; X = victim delayed")
for h in (1, 2, 3):
    print("helper", h)
    for i in range(n):
        row = []
        for j in range(n):
            if i == j or h in (i, j):
                row.append("-")
                continue
            dt = min(ops.wait_blocks(streams[j], streams[i], streams[h], cyc, dev)[0] for _ in range(2))
            row.append("X" if dt > 1.6 * alone else ".")
        print(f"{i:2d} " + " ".join(row))
