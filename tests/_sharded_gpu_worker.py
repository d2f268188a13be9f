"""Worker for tests/test_gpu_sharded.py: one process per GPU (here: however many the launcher started),
both flag transports of the range-sharded search."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
    total = (64 << 20) + 12345
    needle = bytes(range(200, 216))
    for backend in ("torch", "rccl"):
        sh = ss.ShardedSearcher(needle, backend=backend)
        b, e = sh.shard_range(total)
        shard = torch.empty(e - b, dtype=torch.uint8, device="cuda")
        ss.fill_random_device(shard, 0x5EED0001, b)
        assert sh.search_in(shard) is False, backend
        assert sh.find(shard, b) is None
        # plant in the LAST rank's shard only; every rank must see True
        at = total - 16
        if b <= at and at + 16 <= e:
            shard[at - b:at - b + 16] = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
        assert sh.search_in(shard) is True, backend
        assert sh.find(shard, b) == at
        sh.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("sharded gpu worker ok")


if __name__ == "__main__":
    main()
