"""The resident search service (ss_service_*): the same booleans as the launch path and the oracle, the lease and the cap that
bound its residency, stopping it under traffic.  Every wait in the service is bounded on both sides; the tests carry a timeout all
the same.  Tests that read the service's counters (ss_service_counters: a hooks-build entry point) take the `hooks` fixture and
run against libsliceslice_hip_tuning.so - the same host code plus the hooks; the others take the `service` fixture and run against
libsliceslice_hip_service.so (the product's objects plus the service: include/sliceslice_hip_service.h).  The product library
itself holds no service: test_the_product_library_has_no_service."""
import os
import random
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

from conftest import timing_log

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


@pytest.fixture
def hooks(ss):
    with ss.tuning_build():
        yield


@pytest.fixture
def service(ss):
    with ss.service_build():
        yield


def test_the_product_library_has_no_service(ss):
    """The drop-in library is the hot path's surface: the resident service ships apart (libsliceslice_hip_service.so)."""
    assert not ss.lib().has_service and not hasattr(ss.lib(), "ss_service_start")
    with pytest.raises(ss.SlicesliceError) as e:
        ss.SearchService()
    assert "libsliceslice_hip_service.so" in str(e.value)
    with ss.service_build() as L:
        assert L.has_service and not L.has_hooks


def test_service_answers_like_the_launch_path_and_the_oracle(ss, O, hooks):
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
        requests, launches, _ = sv.counters()
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


def test_service_sees_haystack_bytes_written_between_requests(ss, service):
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


def test_bound_haystack_skips_the_acquire_and_answers_the_same(ss, O, hooks):
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
        assert sv.counters()[2] == 0
        got = [sv.search_in(s, t) for s in searchers]
        assert got == [w in raw for w in sample]
        # the first request acquired the range (the needles were uploaded when the searchers were built); a request that starts a
        # new residency (the lease ran out while Python was busy) acquires by itself
        n1 = sv.counters()[2]
        assert len(sample) // 2 < n1 < len(sample)
        got = [sv.search_in(s, t[5:-7]) for s in searchers]   # sub-ranges of the bound range count as bound
        assert got == [w in raw[5:-7] for w in sample]
        n2 = sv.counters()[2]
        assert n1 + len(sample) - 3 <= n2 <= n1 + len(sample)
        late = ss.DynamicHipSearcher.new(b"descriptor")       # uploaded AFTER the latest acquire: its first request is not settled
        assert sv.search_in(late, t) is True and sv.counters()[2] == n2
        assert sv.search_in(late, t) is True and sv.counters()[2] == n2 + 1
        other = torch.zeros(4096, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        before = sv.counters()[2]
        assert sv.search_in(late, other) is False and sv.counters()[2] == before      # outside the range: acquires
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
        assert sv.counters()[2] == before
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


@pytest.mark.timing
def test_service_lease_bounds_the_residency(ss, hooks):
    """Without requests the kernel leaves after its lease, so a device-wide wait cannot hang on it; the next request starts a new
    residency (one more launch) and is answered like any other.  Noise model (profiles/r05/timing_test_spread.jsonl, 10 runs): 5
    launches for the 21 requests every time (bounds 4 .. 21); the device-wide wait returned in under 50 ms (bound 1 s)."""
    t = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    t[-3:] = torch.tensor([7, 8, 9], dtype=torch.uint8)
    torch.cuda.synchronize()
    s = ss.DynamicHipSearcher.new(bytes([7, 8, 9]))
    with ss.SearchService(workgroups=16, lease_ms=2.0) as sv:
        assert sv.search_in(s, t) is True
        assert sv.counters()[:2] == (1, 1)
        t0 = time.perf_counter()
        torch.cuda.synchronize()                             # waits for the service kernel too: at most the lease
        assert time.perf_counter() - t0 < 1.0
        time.sleep(0.05)
        for it in range(20):
            assert sv.search_in(s, t) is True
            if it % 5 == 4:
                time.sleep(0.02)                             # > lease: the kernel has left again
        requests, launches, _ = sv.counters()
        timing_log("service_lease", launches_of_21_requests=launches)
        assert requests == 21 and 4 <= launches <= 21, (requests, launches)
        # a burst shares one residency
        before = sv.counters()[1]
        for _ in range(500):
            assert sv.search_in(s, t) is True
        assert sv.counters()[1] - before <= 5                # (a scheduler hiccup of 2 ms ends a residency)


@pytest.mark.timing
def test_service_residency_is_capped_under_continuous_traffic(ss, hooks):
    """Every request renews the lease, so a caller that never pauses would keep the kernel resident for good - and block every
    device-wide wait in the process with it.  A residency therefore ends after 16 leases (at least 250 ms) whatever the traffic;
    the request that meets the leaving kernel starts the next one.  Two seconds of back-to-back requests: several residencies,
    every answer right, and a device-wide wait from another thread returns while the traffic goes on.  (A capped residency ends
    right BEHIND a request, whose answer may still be on its way when the host sees the kernel gone: the host must not post that
    request a second time - the first cut of this did, and every ~250 ms one answer came back wrong.)  Noise model
    (profiles/r05/timing_test_spread.jsonl, 10 runs): 8-12 launches in the two seconds (bounds 4 .. 40), longest device-wide wait
    0.05-0.2 s (bound 1 s)."""
    import threading
    t = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    t[-3:] = torch.tensor([7, 8, 9], dtype=torch.uint8)
    torch.cuda.synchronize()
    yes, no = ss.DynamicHipSearcher.new(bytes([7, 8, 9])), ss.DynamicHipSearcher.new(bytes([7, 8, 8]))
    waits = []

    def waiter():
        torch.cuda.set_device(0)
        time.sleep(0.3)
        for _ in range(3):
            t0 = time.perf_counter()
            torch.cuda.synchronize()                         # waits for the service kernel too: at most one capped residency
            waits.append(time.perf_counter() - t0)
            time.sleep(0.05)

    with ss.SearchService(workgroups=16, lease_ms=1.0) as sv:         # cap: max(16 leases, 250 ms) = 250 ms
        th = threading.Thread(target=waiter)
        th.start()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.0:
            assert sv.search_in(yes, t) is True and sv.search_in(no, t) is False
            n += 2
        th.join(timeout=30)
        assert not th.is_alive()
        requests, launches, _ = sv.counters()
        timing_log("service_residency_cap", launches_in_2s=launches, longest_device_wait_s=round(max(waits), 3) if waits else None)
        assert requests == n and 4 <= launches <= 40, (requests, launches)      # ~2 s / 250 ms, not one residency and not one per request
        assert len(waits) == 3 and max(waits) < 1.0, waits


@pytest.mark.timing
def test_building_searchers_does_not_wait_for_a_resident_service(ss, hooks):
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
        timing_log("searchers_beside_a_service", forty_constructions_and_searches_s=round(time.perf_counter() - t0, 3))
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


def test_service_restarts_under_a_lease_as_short_as_a_search(ss, hooks):
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
        requests, launches, _ = sv.counters()
        assert requests == 4000 and launches > 100, (requests, launches)


def test_service_is_shared_by_threads(ss, hooks):
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
