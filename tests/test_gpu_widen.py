"""Rows widened in round 3: ss_find_batched - the leftmost offset per problem, the `Option<usize>` shape of the
reference's bench competitors (bench/sse4-strstr/src/lib.rs:4-15) for many problems per call - and ss_search_host_all - the literal
`search_in(&[u8])` (src/x86.rs:523) for a host slice striped over several GPUs' PCIe links."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ss():
    import sliceslice_rs_amd as m
    m.lib()
    return m


def _dev(b):
    return torch.from_numpy(np.frombuffer(bytes(b), dtype=np.uint8).copy()).cuda() if len(b) else torch.empty(0, dtype=torch.uint8, device="cuda")


def test_find_batched_vs_python(ss):
    rng = random.Random(11)
    for round_ in range(4):
        hays, needles = [], []
        for p in range(300):
            kind = rng.random()
            ln = rng.choice([0, 1, 5, 16, 17, 100, 1023, 1024, 4097, 16384, 16385, 70000, 300000]) if kind < 0.8 else rng.randrange(1, 50000)
            alphabet = rng.choice([256, 4, 2])
            h = bytes(rng.randrange(alphabet) for _ in range(min(ln, 3000))) * (ln // 3000 + 1)
            h = h[:ln]
            n = rng.choice([0, 1, 2, 3, 8, 15, 16, 17, 31, 33, 100])
            if ln >= n and n and rng.random() < 0.6:
                at = rng.choice([0, ln - n, rng.randrange(ln - n + 1)])
                nd = h[at:at + n]
            else:
                nd = bytes(rng.randrange(alphabet) for _ in range(n))
            hays.append(h)
            needles.append(nd)
        hb = np.zeros(len(hays) + 1, dtype=np.int64)
        hb[1:] = np.cumsum([len(h) for h in hays])
        nb = np.zeros(len(needles) + 1, dtype=np.int64)
        nb[1:] = np.cumsum([len(n) for n in needles])
        blob, nblob = _dev(b"".join(hays) + b"\0"), _dev(b"".join(needles) + b"\0")
        pos = ss.find_batched(blob, torch.from_numpy(hb).cuda(), nblob, torch.from_numpy(nb).cuda()).cpu().numpy()
        flags = ss.search_batched(blob, torch.from_numpy(hb).cuda(), nblob, torch.from_numpy(nb).cuda()).cpu().numpy()
        for i, (h, nd) in enumerate(zip(hays, needles)):
            want = h.find(nd)
            assert pos[i] == want, (round_, i, len(h), len(nd), int(pos[i]), want)
            assert flags[i] == (1 if want >= 0 else 0), (round_, i)


def test_find_batched_leftmost_across_slices_and_aliased_ranges(ss):
    """Few problems, many slices each (the round-robin layout) and many needles against ONE haystack (aliased ranges, slice-major):
    the leftmost of several occurrences wins wherever the workgroups that see them run."""
    each, count = 4 << 20, 16
    blob = torch.empty(each * count, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(blob, 0xBEEF)
    needle = bytes(range(100, 116))
    pn = _dev(needle)
    want = []
    rng = random.Random(5)
    for i in range(count):
        spots = sorted(rng.sample(range(0, each - 16, 4099), rng.choice([0, 1, 2, 5])))
        if i == 3:
            spots = [16 * 1024 - 8]                       # straddles the first tile edge
        if i == 4:
            spots = [each - 16]
        for sp in spots:
            blob[i * each + sp:i * each + sp + 16] = pn
        want.append(spots[0] if spots else -1)
    torch.cuda.synchronize()
    hay_off = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
    nd_off = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
    nblob = _dev(needle * count)
    for _ in range(3):
        pos = ss.find_batched(blob, hay_off, nblob, nd_off).cpu().tolist()
        assert pos == want
    # the reference's long-haystack loop (bench/benches/i386.rs:246-256) with offsets: every word against the one text
    gd = os.path.join(ROOT, "tests", "golden", "data")
    raw = open(os.path.join(gd, "i386.txt"), "rb").read()
    words = [w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w]
    i386 = _dev(raw)
    lens = np.array([len(w) for w in words], dtype=np.int64)
    nbeg = np.zeros(len(words), dtype=np.int64)
    nbeg[1:] = np.cumsum(lens)[:-1]
    wb = _dev(b"".join(words))
    hb = torch.zeros(len(words), dtype=torch.int64, device="cuda")
    he = torch.full((len(words),), len(raw), dtype=torch.int64, device="cuda")
    pos = ss.find_batched(i386, None, wb, None, hay_ranges=(hb, he), needle_ranges=(torch.from_numpy(nbeg).cuda(), torch.from_numpy(nbeg + lens).cuda())).cpu().numpy()
    for k in range(0, len(words), 7):
        assert pos[k] == raw.find(words[k]), (k, words[k])
    assert (pos >= 0).all()                               # tests/i386.rs:61-70: every word occurs


def test_search_host_all_stripes_a_host_slice_over_devices(ss):
    """One GPU here: the device list names it three and eight times (each range then has its own staging set and thread), plus every
    visible device once.  Matches are planted across every range edge; the boolean must be that of a single search."""
    n_dev = torch.cuda.device_count()
    total = (40 << 20) + 4321
    host = ss.fill_random_host(total, 0x5EED0001).copy()
    needle = bytes(range(200, 233))                        # 33 bytes: ranges overlap by 32
    n = len(needle)
    s = ss.DynamicHipSearcher.new(needle)
    for devices in ([0] * 3, [0] * 8, list(range(n_dev))):
        G = len(devices)
        S = -(-total // G)
        assert ss.search_host_all(s, host, devices) is False
        spots = [0, total - n, total // 2] + [r * S - k for r in range(1, G) for k in (1, n // 2, n - 1)] + [r * S for r in range(1, G)]
        for at in spots:
            saved = host[at:at + n].copy()
            host[at:at + n] = np.frombuffer(needle, dtype=np.uint8)
            assert ss.search_host_all(s, host, devices) is True, (G, at)
            host[at:at + n] = saved
        assert ss.search_host_all(s, host, devices) is False
    assert ss.search_host_all(ss.DynamicHipSearcher.new(b""), host[:10], [0, 0]) is True
    assert ss.search_host_all(s, host[:10], [0, 0]) is False
    with pytest.raises(ss.SlicesliceError):
        ss.search_host_all(s, host, [n_dev + 5])


def test_search_host_all_stops_the_other_devices_at_the_first_match(ss):
    """The threads of ss_search_host_all share one word: the device whose range holds the needle says so, and the others stop
    issuing chunks.  1.5 GiB host slice as three ranges on one GPU, the needle at the start of the first range: far quicker than
    the absent search, which uploads everything."""
    import time
    total = (3 << 29) + 12345
    host = ss.fill_random_host(total, 0x5EED0200).copy()
    needle = bytes(range(180, 200))
    s = ss.DynamicHipSearcher.new(needle)
    devices = [0, 0, 0]
    assert ss.search_host_all(s, host, devices) is False          # (also warms the staging buffers)
    t0 = time.perf_counter()
    assert ss.search_host_all(s, host, devices) is False
    absent = time.perf_counter() - t0
    host[5000:5000 + len(needle)] = np.frombuffer(needle, dtype=np.uint8)
    t0 = time.perf_counter()
    assert ss.search_host_all(s, host, devices) is True
    present = time.perf_counter() - t0
    assert present < 0.5 * absent, (present, absent)
    # the needle in the LAST range only: still found
    host[5000:5000 + len(needle)] = 0
    host[total - 30:total - 30 + len(needle)] = np.frombuffer(needle, dtype=np.uint8)
    assert ss.search_host_all(s, host, devices) is True
