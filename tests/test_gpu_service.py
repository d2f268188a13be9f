"""The resident search service (ss_service_*): the same booleans as the launch path and the oracle, the lease that bounds its
residency, the routing of ss_search_device through it.  Every wait in the service is bounded on both sides; the tests carry
a timeout all the same."""
import os
import random
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ss():
    import sliceslice_rs_amd as m
    m.lib()
    return m


@pytest.fixture(scope="module")
def O():
    from oracle import oracle as m
    return m


def test_service_answers_like_the_launch_path_and_the_oracle(ss, O):
    rng = random.Random(7)
    ln = (4 << 20) + 333
    buf = torch.empty(ln + 32, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(buf, 0xC0FFEE)
    torch.cuda.synchronize()
    with ss.SearchService() as sv:
        for mis in (0, 1, 7, 15):
            t = buf[mis:mis + ln]
            host = t.cpu().numpy()
            hb = host.tobytes()
            cases = [b"", b"\xff", hb[:1], hb[ln - 1:], hb[:16], hb[ln - 16:], hb[12345:12345 + 7], hb[ln // 2:ln // 2 + 33], hb[100:100 + 1500]]
            for n in (2, 3, 5, 8, 13, 16, 17, 31, 64, 200):
                at = rng.randrange(ln - n)
                cases.append(hb[at:at + n])
                miss = bytearray(hb[at:at + n])
                miss[rng.randrange(n)] = 0xFF                   # 0xFF never occurs in the generated haystack
                cases.append(bytes(miss))
            for nd in cases:
                s = ss.DynamicHipSearcher.new(nd)
                want = O.OracleSearcher(nd).search_in(host)
                assert sv.search_in(s, t) == want, (mis, len(nd), "service")
                assert s.search_in(t) == want, (mis, len(nd), "launch path")
            # every prefix length around the tile / piece edges, needle at the very end of the slice
            for cut in (1, 15, 16, 17, 1023, 1024, 1025, 16383, 16384, 16385, 65536 + 5, ln):
                sl = t[:cut]
                for nd in (hb[max(0, cut - 16):cut], hb[:min(cut, 3)], b"\xff\x01"):
                    s = ss.DynamicHipSearcher.new(nd)
                    assert sv.search_in(s, sl) == O.OracleSearcher(nd).search_in(host[:cut]), (mis, cut, len(nd))
        requests, launches = sv.counters()
        # (every searcher construction in the loop above allocates device memory, which waits for the device - i.e. for the
        # service's lease to run out - so here most requests start a residency of their own; bursts are checked below)
        assert requests > 200 and 1 <= launches <= requests
        # with_position and every position of a short needle
        t = buf[:ln]
        host = t.cpu().numpy()
        nd = host[5000:5012].tobytes()
        for pos in range(len(nd)):
            s = ss.DynamicHipSearcher.with_position(nd, pos)
            assert sv.search_in(s, t) is True
        # a filter pair 16 or more apart belongs to the launch path
        long_nd = host[777:777 + 100].tobytes()
        s = ss.DynamicHipSearcher.new(long_nd)
        s.set_filter(0, 99)
        with pytest.raises(ss.SlicesliceError) as e:
            sv.search_in(s, t)
        assert e.value.code == ss.SS_ERR_ARGUMENT


def test_service_sees_haystack_bytes_written_between_requests(ss):
    """The kernel stays resident across requests, so nothing is invalidated for it by a kernel boundary: bytes written to the
    haystack between two requests (by a copy that has completed) must be seen by the next request all the same."""
    ln = 1 << 20
    t = torch.zeros(ln, dtype=torch.uint8, device="cuda")
    needle = bytes(range(1, 17))
    pn = torch.from_numpy(np.frombuffer(needle, dtype=np.uint8).copy()).cuda()
    s = ss.DynamicHipSearcher.new(needle)
    rng = random.Random(3)
    with ss.SearchService(workgroups=32) as sv:
        assert sv.search_in(s, t) is False
        for it in range(200):
            at = rng.choice([0, ln - 16, rng.randrange(ln - 16)])
            t[at:at + 16] = pn
            torch.cuda.current_stream().synchronize()
            assert sv.search_in(s, t) is True, (it, at)
            t[at:at + 16] = 0
            torch.cuda.current_stream().synchronize()
            assert sv.search_in(s, t) is False, (it, at)


def test_bound_haystack_skips_the_acquire_and_answers_the_same(ss, O):
    """ss_service_bind: the caller vouches for a range; requests inside it skip the cache acquire - all but the first, and those
    whose needle reached device memory after the latest acquire.  The answers are the launch path's and the oracle's; after
    unbind, bytes written between requests are seen again."""
    rng = random.Random(11)
    raw = open(os.path.join(ROOT, "tests", "golden", "data", "i386.txt"), "rb").read()
    words = [w for w in open(os.path.join(ROOT, "tests", "golden", "data", "words.txt"), "rb").read().split(b"\n") if w]
    t = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    sample = rng.sample(words, 300) + [b"no such phrase in the manual", b"\x00\x01", raw[-40:], raw[:3], raw[1000:1300]]
    searchers = [ss.DynamicHipSearcher.new(w) for w in sample]
    for s in searchers[:8]:
        s.search_in(t)                                        # (the launch path uploads the needle)
    with ss.SearchService() as sv:
        sv.bind(t)
        assert sv.settled_requests() == 0
        got = [sv.search_in(s, t) for s in searchers]
        assert got == [w in raw for w in sample]
        # the first request acquired the range (the needles were uploaded when the searchers were built); a request that starts a
        # new residency (the lease ran out while Python was busy) acquires by itself
        n1 = sv.settled_requests()
        assert len(sample) // 2 < n1 < len(sample)
        got = [sv.search_in(s, t[5:-7]) for s in searchers]   # sub-ranges of the bound range count as bound
        assert got == [w in raw[5:-7] for w in sample]
        n2 = sv.settled_requests()
        assert n1 + len(sample) - 3 <= n2 <= n1 + len(sample)
        late = ss.DynamicHipSearcher.new(b"descriptor")       # uploaded AFTER the latest acquire: its first request is not settled
        assert sv.search_in(late, t) is True and sv.settled_requests() == n2
        assert sv.search_in(late, t) is True and sv.settled_requests() == n2 + 1
        other = torch.zeros(4096, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        before = sv.settled_requests()
        assert sv.search_in(late, other) is False and sv.settled_requests() == before      # outside the range: acquires
        sv.unbind()
        plant = torch.from_numpy(np.frombuffer(b"descriptor", dtype=np.uint8).copy()).cuda()
        for it in range(50):
            at = rng.randrange(4096 - 10)
            other[at:at + 10] = plant
            torch.cuda.current_stream().synchronize()
            assert sv.search_in(late, other) is True, it
            other[at:at + 10] = 0
            torch.cuda.current_stream().synchronize()
            assert sv.search_in(late, other) is False, it
        assert sv.settled_requests() == before
        # re-binding after a write: the first request of the new binding acquires
        t2 = t.clone()
        torch.cuda.synchronize()
        sv.bind(t2)
        assert sv.search_in(late, t2) is True
        sv.bind(None if False else (0, 0))                    # an empty range unbinds
        t2[:] = 0
        torch.cuda.current_stream().synchronize()
        sv.bind(t2)
        assert sv.search_in(late, t2) is False
        assert sv.search_in(late, t2) is False


def test_service_lease_bounds_the_residency(ss):
    """Without requests the kernel leaves after its lease, so a device-wide wait cannot hang on it; the next request starts a new
    residency (one more launch) and is answered like any other."""
    t = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    t[-3:] = torch.tensor([7, 8, 9], dtype=torch.uint8)
    torch.cuda.synchronize()
    s = ss.DynamicHipSearcher.new(bytes([7, 8, 9]))
    with ss.SearchService(workgroups=16, lease_ms=2.0) as sv:
        assert sv.search_in(s, t) is True
        assert sv.counters() == (1, 1)
        t0 = time.perf_counter()
        torch.cuda.synchronize()                             # waits for the service kernel too: at most the lease
        assert time.perf_counter() - t0 < 1.0
        time.sleep(0.05)
        for it in range(20):
            assert sv.search_in(s, t) is True
            if it % 5 == 4:
                time.sleep(0.02)                             # > lease: the kernel has left again
        requests, launches = sv.counters()
        assert requests == 21 and 4 <= launches <= 21, (requests, launches)
        # a burst shares one residency
        before = sv.counters()[1]
        for _ in range(500):
            assert sv.search_in(s, t) is True
        assert sv.counters()[1] - before <= 5                # (a scheduler hiccup of 2 ms ends a residency)


def test_default_service_routes_ss_search_device(ss, O):
    ln = 1 << 20
    t = torch.empty(ln, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(t, 0xFACADE)
    torch.cuda.synchronize()
    host = t.cpu().numpy()
    present, absent = host[4321:4321 + 16].tobytes(), bytes([255] * 16)
    with ss.SearchService() as sv:
        sv.set_default(True)
        sp, sa = ss.DynamicHipSearcher.new(present), ss.DynamicHipSearcher.new(absent)
        # the service is used only when the caller's stream is idle (it cannot be ordered behind pending work)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(50):
                assert sp.search_in(t) is True and sa.search_in(t) is False
        assert sv.counters()[0] == 100
        for _ in range(5):                                   # the default stream: whichever road it takes, the same answers
            assert sp.search_in(t) is True and sa.search_in(t) is False
        assert 100 <= sv.counters()[0] <= 110
        base = sv.counters()[0]
        # what does not qualify takes the launch path and is still right: a long haystack, a wide pair, a timed search
        big = torch.zeros(32 << 20, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        assert sa.search_in(big) is False
        wide = ss.DynamicHipSearcher.new(host[100:200].tobytes())
        wide.set_filter(0, 99)
        assert wide.search_in(t) is True
        sp.set_timing(True)
        assert sp.search_in(t) is True and sp.last_kernel_ms() > 0
        assert sv.counters()[0] == base
        sv.set_default(False)
        assert sa.search_in(t) is False and sv.counters()[0] == base


def test_parity_suites_in_service_mode():
    """SLICESLICE_SERVICE=1: the library starts a service by itself and routes every qualifying ss_search_device call through it.
    The known-answer, boundary and candidate-heavy suites must pass unchanged that way."""
    env = dict(os.environ, SLICESLICE_SERVICE="1", SLICESLICE_SERVICE_LEASE_MS="5")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                          "-k", "kat or boundary or empty or memchr or candidate or beyond or flush or constructor or position"],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_building_searchers_does_not_wait_for_a_resident_service(ss):
    """Searchers take their device memory from slabs and initialise it through the PCIe BAR: `new` makes no runtime call that
    waits for the device, so a resident service (here with a lease of half a second) does not stall it - with one allocation per
    searcher every `new` below waited out the lease."""
    t = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    t[1000:1004] = torch.tensor([9, 8, 7, 6], dtype=torch.uint8)
    torch.cuda.synchronize()
    warm = [ss.DynamicHipSearcher.new(bytes([1, 2, 3, k % 256, k // 256])) for k in range(300)]      # (more than one slab's worth)
    with ss.SearchService(lease_ms=500.0) as sv:
        assert sv.search_in(warm[0], t) is False
        t0 = time.perf_counter()
        for k in range(40):
            s = ss.DynamicHipSearcher.new(bytes([9, 8, 7, 6]) if k % 2 else bytes([9, 8, 7, 5, k]))
            assert sv.search_in(s, t) is bool(k % 2), k
            del s
        assert time.perf_counter() - t0 < 5.0
        assert sv.counters()[1] == 1                       # one residency throughout


def test_control_blocks_without_bar_writes():
    """SLICESLICE_NO_BAR_WRITES=1: the control block and the needle travel by hipMemcpy (what a platform without a large BAR
    gets); same answers."""
    code = r'''
import os, random, sys
import numpy as np, torch
sys.path.insert(0, %r)
import sliceslice_rs_amd as ss
raw = open(os.path.join(%r, "tests", "golden", "data", "i386.txt"), "rb").read()[:300000]
t = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
rng = random.Random(5)
for _ in range(300):
    n = rng.choice([1, 2, 3, 7, 16, 17, 40, 300, 2500])
    at = rng.randrange(len(raw) - n)
    nd = bytearray(raw[at:at + n])
    if rng.random() < 0.4:
        nd[rng.randrange(n)] ^= 0x80
    nd = bytes(nd)
    s = ss.DynamicHipSearcher.new(nd)
    want = raw.find(nd)
    assert s.search_in(t) is (want >= 0), (nd, want)
    assert s.find(t) == (want if want >= 0 else None), (nd, want)
print("ok")
''' % (ROOT, ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, SLICESLICE_NO_BAR_WRITES="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_service_restarts_under_a_lease_as_short_as_a_search(ss):
    """The lease at its minimum (50 us): the kernel leaves between almost any two requests, and requests keep arriving while the
    stop word spreads - taken by some waves and not by others, never completed, posted again to a new residency.  Every answer
    must still be right."""
    rng = random.Random(23)
    t = torch.zeros(300000, dtype=torch.uint8, device="cuda")
    t[200000:200005] = torch.tensor([5, 4, 3, 2, 1], dtype=torch.uint8)
    torch.cuda.synchronize()
    yes, no = ss.DynamicHipSearcher.new(bytes([5, 4, 3, 2, 1])), ss.DynamicHipSearcher.new(bytes([5, 4, 3, 2, 2]))
    small = t[199990:200020]
    with ss.SearchService(workgroups=48, lease_ms=0.05) as sv:
        for it in range(4000):
            s, want = (yes, True) if rng.random() < 0.5 else (no, False)
            hay = small if rng.random() < 0.3 else t
            assert sv.search_in(s, hay) is want, it
            r = rng.random()
            if r < 0.3:
                time.sleep(rng.random() * 2e-4)
            elif r < 0.5:
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < rng.random() * 1e-4:
                    pass
        requests, launches = sv.counters()
        assert requests == 4000 and launches > 100, (requests, launches)


def test_service_is_shared_by_threads(ss):
    """One request at a time per service: callers from several threads queue on the service's mutex and each gets ITS answer
    (ctypes releases the GIL, so the calls do overlap in the library)."""
    import threading
    rng = random.Random(31)
    raw = open(os.path.join(ROOT, "tests", "golden", "data", "i386.txt"), "rb").read()[:400000]
    t = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    needles = []
    for _ in range(64):
        n = rng.choice([2, 5, 9, 16, 17, 33])
        at = rng.randrange(len(raw) - n)
        nd = bytearray(raw[at:at + n])
        if rng.random() < 0.5:
            nd[rng.randrange(n)] ^= 0x40
        needles.append(bytes(nd))
    searchers = [ss.DynamicHipSearcher.new(nd) for nd in needles]
    want = [nd in raw for nd in needles]
    wrong = []
    with ss.SearchService() as sv:
        sv.bind(t)

        def worker(k):
            torch.cuda.set_device(0)
            for rep in range(300):
                i = (k * 17 + rep * 5) % len(searchers)
                if sv.search_in(searchers[i], t) is not want[i]:
                    wrong.append((k, rep, i))

        threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not wrong, wrong[:5]
        assert sv.counters()[0] == 4 * 300
