"""Can two RCCL ranks share ONE GPU (so that the nccl code path of parallel.py could be exercised on a one-GPU box)?
Spawns two ranks on device 0 and tries an all-reduce; prints what happened."""
import os, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp


def run(rank, ws):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=ws, device_id=torch.device("cuda", 0))
        t = torch.full((4,), float(rank + 1), device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce -> {t.tolist()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print(f"rank {rank}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)


if __name__ == "__main__":
    mp.spawn(run, args=(2,), nprocs=2, join=True)
