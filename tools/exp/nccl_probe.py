"""Can RCCL run two ranks on ONE GPU here?  (tools/exp; prints the outcome)"""
import os, sys, socket
import torch.multiprocessing as mp

def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        t = torch.full((4,), float(rank + 1), device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        q.put((rank, "ok", t.cpu().tolist()))
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, "fail", repr(e)[:300]))

if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps: p.start()
    for _ in ps:
        try: print(q.get(timeout=90))
        except Exception as e: print("timeout", e)
    for p in ps:
        p.join(10)
        if p.is_alive(): p.kill()
