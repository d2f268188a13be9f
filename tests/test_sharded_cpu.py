"""world_size-2 gloo test of the range-sharded search: shard ranges with n-1 overlap and
the single all-reduce(MAX) of the found flag.  The per-shard scan is injected (the CPU oracle) because
the product has no CPU search path; what is under test is the product's sharding + combine logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sliceslice_rs_amd as ss
        from oracle import oracle as O
        logical = ss.fill_random_host(total, 0x5EED0001).copy()       # same bytes on every rank
        S = -(-total // world)
        results = []

        def run(needle, hay):
            sh = ss.ShardedSearcher(needle, local_search=lambda shard: O.OracleSearcher(needle).search_in(shard))
            b, e = sh.shard_range(len(hay))
            got = sh.search_in(hay[b:e])
            want = O.naive_contains(hay, needle)
            return (got, want, (b, e))

        n16 = bytes(range(200, 216))
        cases = []
        cases.append(("absent", n16, logical))
        for name, at in (("rank0_only", 10), ("rank1_only", total - 16), ("straddle_mid", S - 8),
                         ("straddle_1byte_left", S - 1), ("straddle_15_left", S - 15), ("starts_at_boundary", S)):
            h = logical.copy()
            h[at:at + 16] = np.frombuffer(n16, dtype=np.uint8)
            cases.append((name, n16, h))
        big = bytes((7 * k + 3) % 251 for k in range(S + 10))          # needle longer than a shard's own part
        h = logical.copy()
        h[5:5 + len(big)] = np.frombuffer(big, dtype=np.uint8)
        cases.append(("needle_longer_than_shard", big, h))
        cases.append(("one_byte_absent", b"\xff", logical))
        cases.append(("one_byte_present", logical[total - 1:].tobytes(), logical))
        cases.append(("empty_needle", b"", logical))
        for name, needle, hay in cases:
            got, want, rng = run(needle, hay)
            results.append((name, got, want, rng))
            # sharded find: global leftmost offset through one all-reduce(MIN)
            if len(needle) > 0:
                def local_find(shard, nd=needle):
                    p = shard.tobytes().find(nd)
                    return None if p < 0 else p
                sh = ss.ShardedSearcher(needle, local_find=local_find)
                b, e = sh.shard_range(len(hay))
                gotp = sh.find(hay[b:e], b)
                wantp = hay.tobytes().find(needle)
                results.append((name + ":find", gotp, None if wantp < 0 else wantp, (b, e)))
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_gloo(world):
    total = 100_003
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    [p.start() for p in procs]
    out = dict(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    names = [r[0] for r in out[0]]
    assert "straddle_mid" in names
    for rank in range(world):
        for name, got, want, rng in out[rank]:
            assert got == want, (rank, name, rng)
    # every rank returns the same (combined) boolean
    for i in range(len(names)):
        assert len({out[r][i][1] for r in range(world)}) == 1


def _rccl_refusal_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SLICESLICE_RCCL_INIT_TIMEOUT"] = "20"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sliceslice_rs_amd as ss
        try:
            ss.ShardedSearcher(b"needle", local_search=lambda shard: False, backend="rccl")
            q.put((rank, "built"))
        except ss.SlicesliceError as e:
            q.put((rank, "refused: %s" % e))
        dist.barrier()                                # both ranks are still in step afterwards
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_native_rccl_init_fails_on_all_ranks_together():
    """No GPU here, so the native communicator cannot be built: every rank must get the SAME SlicesliceError (bench.py
    then moves the flag with torch.distributed instead) - none may be left waiting in a broadcast or inside
    ncclCommInitRank for a rank that has already given up."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the communicator may well come up")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_refusal_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive, "a rank hung in the native RCCL bootstrap"
    got = dict(q.get(timeout=5) for _ in range(world))
    assert all(v.startswith("refused") for v in got.values()), got


def _failing_rank_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sliceslice_rs_amd as ss
        hay = np.frombuffer(b"x" * 1000 + b"needle" + b"y" * 1000, dtype=np.uint8)
        calls = {"n": 0}

        def local_search(shard):
            calls["n"] += 1
            if calls["n"] == 2 and rank == 1:                     # the second search fails on rank 1 only
                raise RuntimeError("injected local failure on rank 1")
            return b"needle" in shard.tobytes()

        def local_find(shard):
            calls["n"] += 1
            if calls["n"] == 5 and rank == 1:
                raise RuntimeError("injected local find failure on rank 1")
            p = shard.tobytes().find(b"needle")
            return None if p < 0 else p
        sh = ss.ShardedSearcher(b"needle", local_search=local_search, local_find=local_find)
        b, e = sh.shard_range(len(hay))
        out = []
        for step in range(3):                                     # ok, failure, ok again: the ranks stay in step
            try:
                out.append(("ok", sh.search_in(hay[b:e])))
            except ss.SlicesliceError as exc:
                out.append(("peer", exc.code))
            except RuntimeError as exc:
                out.append(("local", str(exc)))
        for step in range(3):                                     # the same for find(): calls 4, 5, 6
            try:
                out.append(("ok", sh.find(hay[b:e], b)))
            except ss.SlicesliceError as exc:
                out.append(("peer", exc.code))
            except RuntimeError as exc:
                out.append(("local", str(exc)))
        dist.barrier()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 3])
def test_a_failing_rank_leaves_nobody_in_the_collective(world):
    """a rank-local failure must not keep that rank out of the all-reduce.  The failing rank still
    contributes ("not found" + the pair's second word), then raises its own error; every other rank raises SS_ERR_PEER;
    the next search works again on all of them.  (Native form: ss_search_sharded / ss_find_sharded do the same with a
    two-int all-reduce - tests/test_gpu_sharded.py injects the failure there with ss_debug_fail_next_scans.)"""
    import sliceslice_rs_amd as ss
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive, "a rank hung: somebody skipped the collective"
    got = dict(q.get(timeout=5) for _ in range(world))
    for rank in range(world):
        out = got[rank]
        assert out[0] == ("ok", True) and out[2] == ("ok", True), (rank, out)
        assert out[3] == ("ok", 1000) and out[5] == ("ok", 1000), (rank, out)
        if rank == 1:
            assert out[1][0] == "local" and out[4][0] == "local", out
        else:
            assert out[1] == ("peer", ss.SS_ERR_PEER) and out[4] == ("peer", ss.SS_ERR_PEER), (rank, out)
