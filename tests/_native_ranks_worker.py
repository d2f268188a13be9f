"""Worker of tests/test_gpu_sharded.py: the library's NATIVE collective code (ss_comm_init_rank / ss_search_sharded /
ss_find_sharded; ss_comm_init_all / ss_search_sharded_all / ss_find_sharded_all with the RCCL combine) with 2, 3, 8 ranks on ONE
GPU, through the shared-memory RCCL stand-in (tests/native/fake_rccl.c; SLICESLICE_RCCL_LIB is set by the parent test).  Real RCCL
refuses two ranks on one device, so before this the C code of the collective paths had only ever run with nranks == 1.  No
torch.distributed anywhere: rank 0 creates the unique id and hands it to the others through a file.

    python _native_ranks_worker.py rank <rank> <nranks> <id-file> [loops]
    python _native_ranks_worker.py set <G>
With SLICESLICE_HIP_LIB pointing at a hooks build (libsliceslice_hip_tuning.so) the rank mode also injects a failure into one
rank's scan (ss_debug_fail_next_scans) and checks that nobody is left waiting in the collective."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sliceslice_rs_amd as ss  # noqa: E402

NPOS = (1 << 64) - 1
SEED = 0x5EED0001


class Rank:
    """One rank's end of a native communicator (no Python wrapper in between: the C ABI as a Rust / C caller would use it)."""

    def __init__(self, rank, nranks, id_file, needle):
        self.L = ss.lib()
        self.rank, self.nranks = rank, nranks
        uid = (ctypes.c_uint8 * 128)()
        if rank == 0:
            assert self.L.ss_comm_unique_id(uid) == 0, self.L.ss_last_error()
            with open(id_file + ".tmp", "wb") as f:
                f.write(bytes(uid))
            os.replace(id_file + ".tmp", id_file)
        else:
            t0 = time.time()
            while not os.path.exists(id_file):
                assert time.time() - t0 < 120, "rank 0 never wrote the unique id"
                time.sleep(0.01)
            uid = (ctypes.c_uint8 * 128).from_buffer_copy(open(id_file, "rb").read())
        self.comm = ctypes.c_void_p()
        rc = self.L.ss_comm_init_rank(uid, nranks, rank, ctypes.byref(self.comm))
        assert rc == 0, self.L.ss_last_error()
        n = ctypes.c_int(0)
        assert self.L.ss_comm_count(self.comm, ctypes.byref(n)) == 0 and n.value == nranks
        self.searcher = ss.DynamicHipSearcher.new(needle)
        self.needle = needle

    def search(self, shard):
        found = ctypes.c_int(-1)
        rc = self.L.ss_search_sharded(self.searcher._h, shard.data_ptr(), shard.numel(), self.comm, 0, ctypes.byref(found))
        return rc, found.value

    def find(self, shard, begin):
        pos = ctypes.c_uint64(12345)
        rc = self.L.ss_find_sharded(self.searcher._h, shard.data_ptr(), shard.numel(), begin, self.comm, 0, ctypes.byref(pos))
        return rc, (None if pos.value == NPOS else pos.value)

    def close(self):
        self.L.ss_comm_free(self.comm)


def rank_main(rank, nranks, id_file, loops):
    torch.cuda.set_device(0)
    needle = bytes(range(200, 216))
    n = len(needle)
    total = (24 << 20) + 12345
    S = -(-total // nranks)
    b, e = ss.shard_range(total, n, nranks, rank)
    assert b == min(rank * S, total) and e == min(total, b + S + n - 1)
    shard = torch.empty(e - b, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(shard, SEED, b)
    pn = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    R = Rank(rank, nranks, id_file, needle)

    def plant(at, with_needle=True, keep=None):
        """This rank's part of needle bytes at global offset `at` (every rank calls it; most hold nothing of it)."""
        lo, hi = max(at, b), min(at + n, e)
        if lo >= hi:
            return None
        saved = shard[lo - b:hi - b].clone() if keep is None else keep
        shard[lo - b:hi - b] = pn[lo - at:hi - at] if with_needle else saved
        torch.cuda.synchronize()
        return saved

    assert R.search(shard) == (0, 0) and R.find(shard, b) == (0, None)
    # a match in the middle of every rank's shard, at both ends of the haystack, straddling every boundary by 1 .. n-1 bytes
    spots = [0, total - n] + [min(r * S + S // 2, total - n) for r in range(nranks)]
    spots += [r * S - k for r in range(1, nranks) for k in (1, n // 2, n - 1)] + [r * S for r in range(1, nranks)]
    for at in spots:
        saved = plant(at)
        assert R.search(shard) == (0, 1), (rank, at)
        assert R.find(shard, b) == (0, at), (rank, at)
        plant(at, False, saved)
        assert R.search(shard) == (0, 0), (rank, at)
    # two occurrences in different shards: the leftmost wins on every rank
    a1, a2 = S // 3, (nranks - 1) * S + 1000
    s1, s2 = plant(a1), plant(a2)
    assert R.find(shard, b) == (0, a1) and R.search(shard) == (0, 1)
    plant(a1, False, s1)
    assert R.find(shard, b) == (0, a2)
    plant(a2, False, s2)
    assert R.find(shard, b) == (0, None)
    # shards shorter than the needle (nothing to scan on any rank), and the empty needle (answered without the collective)
    assert R.L.ss_search_sharded(R.searcher._h, shard.data_ptr(), 5, R.comm, 0, ctypes.byref(ctypes.c_int(0))) == 0
    empty = ss.DynamicHipSearcher.new(b"")
    f = ctypes.c_int(0)
    assert R.L.ss_search_sharded(empty._h, shard.data_ptr(), shard.numel(), R.comm, 0, ctypes.byref(f)) == 0 and f.value == 1

    # back-to-back searches on small shards: the epochs of `loops` calls, the every-256th stream wait, the answer word behind the
    # all-reduce; the needle shows up in one rank's shard now and then (decided from the call index: every rank agrees)
    small = shard[:1 << 20]
    sb = b
    own = (rank * 7919) % (small.numel() - n)
    t0 = time.perf_counter()
    for it in range(loops):
        present_on = it % (3 * nranks) if it % 5 == 0 else -1
        if present_on == rank:
            keep = small[own:own + n].clone()
            small[own:own + n] = pn
            torch.cuda.synchronize()
        rc, found = R.search(small)
        assert rc == 0 and found == (1 if 0 <= present_on < nranks else 0), (rank, it, rc, found)
        if it % 97 == 0:
            rc, pos = R.find(small, sb)
            assert rc == 0, (rank, it, rc)
            if present_on == rank:
                assert pos == sb + own, (rank, it, pos)
            else:
                assert (pos is not None) == (0 <= present_on < nranks), (rank, it, pos)
        if present_on == rank:
            small[own:own + n] = keep
            torch.cuda.synchronize()
    per_call_us = (time.perf_counter() - t0) / max(loops, 1) * 1e6

    failures = 0
    if R.L.has_hooks:
        # A rank-local failure (the LAST rank's scan is never enqueued) must not keep that rank out of the collective: it returns its
        # own error, every other rank SS_ERR_PEER, nobody hangs, and the next search finds all ranks in step.
        saved = plant(total - n)
        for call in ("search", "find", "search"):
            if rank == nranks - 1:
                assert R.L.ss_debug_fail_next_scans(R.searcher._h, 1) == 0
            rc = R.search(shard)[0] if call == "search" else R.find(shard, b)[0]
            assert rc == (ss.SS_ERR_HIP if rank == nranks - 1 else ss.SS_ERR_PEER), (rank, call, rc)
            failures += 1
            assert R.search(shard) == (0, 1) and R.find(shard, b) == (0, total - n), (rank, call)
        plant(total - n, False, saved)
        # the epoch wrap of the communicator's flag pair, crossed by all ranks together
        assert R.L.ss_debug_set_comm_epoch(R.comm, None, 2**31 - 4) == 0
        for it in range(8):
            assert R.search(shard) == (0, 0), it
            saved = plant(total - n)
            assert R.search(shard) == (0, 1), it
            plant(total - n, False, saved)
    late = int(ss.lib().ss_debug_late_answers()) if R.L.has_hooks else -1
    R.close()
    print("rank %d of %d ok: %d spots, %d back-to-back searches at %.1f us each, %d injected failures, %d late answers" %
          (rank, nranks, len(spots), loops, per_call_us, failures, late), flush=True)


def set_main(G):
    """All ranks in ONE process: ss_comm_init_all over device 0 listed G times (the stand-in allows it), the grouped all-reduce of
    ss_search_sharded_all / ss_find_sharded_all against the host combine."""
    torch.cuda.set_device(0)
    needle = bytes(range(100, 133))                      # 33 bytes: shards overlap by 32
    n = len(needle)
    total = (24 << 20) + 4321
    node = ss.NodeSearcher(needle, devices=[0] * G)
    ranges = [node.shard_range(total, g) for g in range(G)]
    S = -(-total // G)
    logical = torch.empty(total, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(logical, SEED)
    pn = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
    begins = [b for b, _ in ranges]

    def shards():
        return [logical[b:e].clone() for b, e in ranges]  # every shard its own allocation, like on G devices
    checked = 0
    assert node.rccl_ranks() == G                        # ncclCommCount of every communicator of the set
    # both combines, each with one issue thread per device (the default from two devices up) and with everything issued from
    # the calling thread (the all-reduces as one group)
    for mode, issue in ((ss.NodeSearcher.COMBINE_RCCL, ss.NodeSearcher.ISSUE_THREADS), (ss.NodeSearcher.COMBINE_HOST, ss.NodeSearcher.ISSUE_THREADS),
                        (ss.NodeSearcher.COMBINE_RCCL, ss.NodeSearcher.ISSUE_SERIAL), (ss.NodeSearcher.COMBINE_HOST, ss.NodeSearcher.ISSUE_SERIAL)):
        node.set_combine(mode)
        node.set_issue(issue)
        assert node.search_in(shards()) is False and node.find(shards(), begins) is None
        spots = [0, total - n, total // 2] + [r * S - k for r in range(1, G) for k in (1, n // 2, n - 1)] + [r * S for r in range(1, G)]
        for at in spots:
            saved = logical[at:at + n].clone()
            logical[at:at + n] = pn
            sh = shards()
            assert node.search_in(sh) is True, (G, mode, at)
            assert node.find(sh, begins) == at, (G, mode, at)
            logical[at:at + n] = saved
            checked += 1
        a1, a2 = S + 100, (G - 1) * S + 5000
        s1, s2 = logical[a1:a1 + n].clone(), logical[a2:a2 + n].clone()
        logical[a1:a1 + n] = pn
        logical[a2:a2 + n] = pn
        assert node.find(shards(), begins) == a1
        logical[a1:a1 + n] = s1
        logical[a2:a2 + n] = s2
        sh = shards()
        for it in range(600):                            # epochs, the every-256th stream wait
            assert node.search_in(sh) is False, it
        tiny = [s[:5] for s in sh]
        assert node.search_in(tiny) is False and node.find(tiny, begins) is None
        node._searcher.set_timing(True)
        assert node.search_in(sh) is False
        ms, us = node.last_kernel_ms(), node.last_issue_us()
        assert len(ms) == G and all(m > 0 for m in ms) and len(us) == 4 and us[3] >= us[0] > 0, (ms, us)
        node._searcher.set_timing(False)
        if node._L.has_hooks:
            # a device whose scan cannot be enqueued: the others still get through their collective (nobody hangs), the call
            # fails, and the next search finds every device in step again
            logical[total - n:] = pn
            sh2 = shards()
            for fails in (1, 2):
                node._L.ss_debug_fail_next_scans(node._searcher._h, fails)
                try:
                    node.search_in(sh2)
                    raise AssertionError("a search with an injected scan failure returned an answer")
                except ss.SlicesliceError as e:
                    assert e.code == ss.SS_ERR_HIP, e
                node._L.ss_debug_fail_next_scans(node._searcher._h, 0)
                assert node.search_in(sh2) is True and node.search_in(sh) is False
            ss.fill_random_device(logical, SEED)
    if node._L.has_hooks:
        node.set_combine(ss.NodeSearcher.COMBINE_RCCL)
        node.set_epoch(2**31 - 3)
        for it in range(4):
            assert node.search_in(shards()) is False
            logical[total - n:] = pn
            assert node.search_in(shards()) is True
            ss.fill_random_device(logical, SEED)
    late = int(ss.lib().ss_debug_late_answers()) if node._L.has_hooks else -1
    node.close()
    print("set of %d ok: %d spots per combine and issue mode, %d late answers" % (G, checked // 4, late), flush=True)


def relay_main():
    """Cross-device early exit of ss_search_sharded_all: a match on ONE device ends the other devices' scans too - the host, which
    waits for the answer words anyway, sees the finding wave's pinned mirror and stores the epoch into the other devices' flags
    through the BAR.  Three shards of 3 GiB on one GPU, the needle at the start of shard 0 only: with the relay the call returns
    long before the other two shards have been read; without it (SLICESLICE_CROSS_EXIT=0, a hooks-build switch) it takes their
    full scans.  The answers do not change."""
    torch.cuda.set_device(0)
    assert ss.lib().has_hooks
    G, each = 3, 3 << 30
    needle = bytes(range(200, 216))
    pn = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
    sh = []
    for g in range(G):
        t = torch.empty(each, dtype=torch.uint8, device="cuda")
        ss.fill_random_device(t, 0x5EED0100 + g)
        sh.append(t)
    sh[0][4096:4096 + 16] = pn
    torch.cuda.synchronize()
    times = {}
    for combine in (ss.NodeSearcher.COMBINE_HOST, ss.NodeSearcher.COMBINE_RCCL):
        for relay in ("1", "0"):
            os.environ["SLICESLICE_CROSS_EXIT"] = relay
            node = ss.NodeSearcher(needle, devices=[0] * G)
            node.set_combine(combine)
            for _ in range(5):
                assert node.search_in(sh) is True
            t0 = time.perf_counter()
            for _ in range(20):
                assert node.search_in(sh) is True
            times[combine, relay] = (time.perf_counter() - t0) / 20
            node.close()
    os.environ.pop("SLICESLICE_CROSS_EXIT")
    sh[0][4096:4096 + 16] = 0                                # absent: nothing to relay, same answer either way
    torch.cuda.synchronize()
    node = ss.NodeSearcher(needle, devices=[0] * G)
    assert node.search_in(sh) is False
    node.close()
    host = ss.NodeSearcher.COMBINE_HOST
    assert times[host, "1"] < 0.6 * times[host, "0"], times
    print("relay ok: %s" % {("host" if c == host else "rccl") + ("+relay" if r == "1" else ""): round(v * 1e3, 3) for (c, r), v in times.items()}, flush=True)


if __name__ == "__main__":
    assert os.environ.get("SLICESLICE_RCCL_LIB"), "the parent test sets SLICESLICE_RCCL_LIB to the stand-in"
    if sys.argv[1] == "rank":
        rank_main(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 2000)
    elif sys.argv[1] == "set":
        set_main(int(sys.argv[2]))
    else:
        relay_main()
