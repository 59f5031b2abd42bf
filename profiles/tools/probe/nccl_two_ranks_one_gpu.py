# probe: can RCCL run two ranks on ONE device?
import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda:0"))
        t = torch.ones(4, device="cuda:0") * (rank + 1)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print("rank", rank, "ok", t.tolist(), flush=True)
    except Exception as e:
        print("rank", rank, "FAILED", type(e).__name__, str(e)[:400], flush=True)
if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=w, args=(r, 29511)) for r in range(2)]
    [p.start() for p in ps]
    for p in ps:
        p.join(90)
        if p.is_alive(): p.kill(); print("timeout-killed")
