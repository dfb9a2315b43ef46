import os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.full((1024,), rank, dtype=torch.uint8, device="cuda")
    out = torch.empty((world * 1024,), dtype=torch.uint8, device="cuda")
    w = dist.all_gather_into_tensor(out, x, async_op=True); w.wait(); torch.cuda.synchronize()
    c = torch.tensor([rank + 1], dtype=torch.int64, device="cuda"); dist.all_reduce(c); torch.cuda.synchronize()
    print("rank", rank, "ok", out[::1024].tolist(), int(c.item()), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank", rank, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
