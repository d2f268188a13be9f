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
    # SS_TEST_SHARE_GPU=1: several ranks share cuda:0 and the flag travels over gloo - exercises the real
    # shard kernels + overlap + combine with world_size > 1 on a one-GPU box (RCCL refuses duplicate GPUs).
    share = os.environ.get("SS_TEST_SHARE_GPU") == "1"
    torch.cuda.set_device(0 if share else int(os.environ.get("LOCAL_RANK", "0")))
    if share:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
    total = (64 << 20) + 12345
    needle = bytes(range(200, 216))
    if share:
        S = -(-total // world)
        sh = ss.ShardedSearcher(needle, backend="torch")
        b, e = sh.shard_range(total)
        assert b == rank * S and e == min(total, b + S + 15)
        shard = torch.empty(e - b, dtype=torch.uint8, device="cuda")
        ss.fill_random_device(shard, 0x5EED0001, b)
        pn = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
        assert sh.search_in(shard) is False and sh.find(shard, b) is None
        # matches straddling every shard boundary by 1..15 bytes, at the very start and the very end
        spots = [0, total - 16] + [r * S - k for r in range(1, world) for k in (1, 8, 15)] + [r * S for r in range(1, world)]
        for at in spots:
            saved = {}
            lo, hi = max(at, b), min(at + 16, e)
            if lo < hi:                                  # this rank holds (part of) the planted bytes
                saved = shard[lo - b:hi - b].clone()
                shard[lo - b:hi - b] = pn[lo - at:hi - at]
            assert sh.search_in(shard) is True, (rank, at)
            assert sh.find(shard, b) == at, (rank, at)
            if lo < hi:
                shard[lo - b:hi - b] = saved
            assert sh.search_in(shard) is False
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            print("sharded gpu worker ok")
        return
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
        # A rank-local failure (the last rank's scan is never enqueued: ss_debug_fail_next_scans) must not keep that rank
        # out of the collective: it raises its own error, every other rank SS_ERR_PEER, nobody hangs, and the next
        # search finds all ranks in step.
        for call in ("search", "find") if ss.lib().has_hooks else ():
            if rank == world - 1:
                sh.fail_next_scans(1)
            try:
                sh.search_in(shard) if call == "search" else sh.find(shard, b)
                outcome = "answered"
            except ss.SlicesliceError as exc:
                outcome = exc.code
            assert outcome == (ss.SS_ERR_HIP if rank == world - 1 else ss.SS_ERR_PEER), (backend, call, rank, outcome)
            assert sh.search_in(shard) is True, (backend, call)
            assert sh.find(shard, b) == at, (backend, call)
        sh.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("sharded gpu worker ok")


if __name__ == "__main__":
    main()
