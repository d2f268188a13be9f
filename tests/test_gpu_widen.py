"""ss_find_batched - the leftmost offset per problem, the `Option<usize>` shape of the reference's bench competitors
(bench/sse4-strstr/src/lib.rs:4-15) for many problems per call - and ss_batch_plan_*: the per-problem set-up done once, searched
many times (the reference builds its searchers once and times the searches, bench/benches/i386.rs:246-256)."""
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




def _batch(ss, rng, count, hay_len, needle_len, present_every=3):
    """`count` problems over one haystack blob (hay_len each) and one needle blob; every `present_every`-th needle is cut out of
    its haystack.  Returns device tensors + the expected leftmost offsets (-1: absent)."""
    hay = torch.empty(count * hay_len, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0xBA7C4)
    host = hay.cpu().numpy()
    needles = bytearray()
    noff = [0]
    want = []
    for i in range(count):
        h = host[i * hay_len:(i + 1) * hay_len].tobytes()
        n = needle_len if isinstance(needle_len, int) else rng.choice(needle_len)
        if i % present_every == 0 and n <= hay_len:
            at = rng.randrange(hay_len - n + 1)
            nd = h[at:at + n]
        else:
            nd = bytes([255]) * max(n, 1)                 # 0xFF never occurs in the generated haystack
        needles += nd
        noff.append(len(needles))
        want.append(h.find(nd) if len(nd) else 0)
    hoff = torch.arange(0, (count + 1) * hay_len, hay_len, dtype=torch.int64, device="cuda")
    nbuf = torch.from_numpy(np.frombuffer(bytes(needles), dtype=np.uint8).copy()).cuda()
    return hay, hoff, nbuf, torch.tensor(noff, dtype=torch.int64, device="cuda"), want


def test_batch_plan_runs_equal_the_unplanned_calls(ss):
    """A plan's run() must give what search_batched / find_batched give on the same problems - on the first run and on every later
    one, with haystack CONTENTS changed between runs (plants added and removed), into output buffers full of garbage."""
    rng = random.Random(41)
    # (1 x 24 MiB and 3 x 9 MiB: long problems - four tiles per workgroup, round robin, hundreds of workgroups per problem)
    for count, hay_len, nlen in ((1, 3 << 20, 16), (7, 1 << 20, (1, 2, 5, 16, 17, 40)), (300, 65536, (3, 16, 33)), (4096, 4096, 16),
                                 (50, 100, (1, 4, 16, 100, 101)), (1, 24 << 20, 16), (3, 9 << 20, (5, 16))):
        hay, hoff, nbuf, noff, want = _batch(ss, rng, count, hay_len, nlen)
        plan_s = ss.BatchPlan(hay, hoff, nbuf, noff)
        plan_f = ss.BatchPlan(hay, hoff, nbuf, noff, find=True)
        out_s = torch.full((count,), 77, dtype=torch.int32, device="cuda")
        out_f = torch.full((count,), 77, dtype=torch.int64, device="cuda")
        for rep in range(4):
            plan_s.run(out_s)
            plan_f.run(out_f)
            assert out_s.tolist() == [1 if w >= 0 else 0 for w in want], (count, hay_len, rep)
            assert out_f.tolist() == want, (count, hay_len, rep)
            assert ss.search_batched(hay, hoff, nbuf, noff).tolist() == out_s.tolist()
            assert ss.find_batched(hay, hoff, nbuf, noff).tolist() == want
            out_s.fill_(-5)
            out_f.fill_(-5)
        # contents change between runs: wipe every planted needle, then put one back further left
        host = hay.cpu().numpy().copy()
        nb = nbuf.cpu().numpy().tobytes()
        no = noff.tolist()
        first = next((i for i, w in enumerate(want) if w >= 0 and no[i + 1] - no[i] >= 3 and hay_len >= 64), None)
        if first is not None:
            nd = nb[no[first]:no[first + 1]]
            base = first * hay_len
            hay[base + want[first]] ^= 0x55                                   # gone
            torch.cuda.synchronize()
            h = hay[base:base + hay_len].cpu().numpy().tobytes()
            w2 = h.find(nd)
            plan_f.run(out_f)
            plan_s.run(out_s)
            assert out_f[first].item() == w2 and out_s[first].item() == (1 if w2 >= 0 else 0)
            hay[base:base + len(nd)] = torch.from_numpy(np.frombuffer(nd, dtype=np.uint8).copy()).cuda()   # back, at offset 0
            torch.cuda.synchronize()
            plan_f.run(out_f)
            plan_s.run(out_s)
            assert out_f[first].item() == 0 and out_s[first].item() == 1
        plan_s.close()
        plan_f.close()


def test_batch_plan_trivial_problems_positions_and_graph_replay(ss):
    """Problems without a scan (empty needle, haystack shorter than the needle, a position that breaks the with_position rules)
    are answered from the plan on every run; and a run is ONE kernel launch with nothing allocated, so it can be captured into a
    hipGraph and replayed - which the unplanned call refuses (its per-stream scratch may be reallocated by a later call)."""
    hay = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    hay[100:103] = torch.tensor([1, 2, 3], dtype=torch.uint8)
    needles = torch.tensor([1, 2, 3, 9, 9, 9, 9, 9], dtype=torch.uint8, device="cuda")
    # problems: [1,2,3] in hay (found); [] (empty needle: 1); [9]*5 in a 3-byte haystack (0); [1,2,3] with position 3 (bad); [9,9] absent
    hb = torch.tensor([0, 0, 100, 0, 0], dtype=torch.int64, device="cuda")
    he = torch.tensor([4096, 4096, 103, 4096, 4096], dtype=torch.int64, device="cuda")
    nb = torch.tensor([0, 3, 3, 0, 3], dtype=torch.int64, device="cuda")
    ne = torch.tensor([3, 3, 8, 3, 5], dtype=torch.int64, device="cuda")
    pos = torch.tensor([2, 7, 4, 3, 1], dtype=torch.int64, device="cuda")
    plan = ss.BatchPlan(hay, None, needles, None, position=pos, hay_ranges=(hb, he), needle_ranges=(nb, ne))
    fplan = ss.BatchPlan(hay, None, needles, None, find=True, hay_ranges=(hb, he), needle_ranges=(nb, ne))
    out = torch.full((5,), 9, dtype=torch.int32, device="cuda")
    fout = torch.full((5,), 9, dtype=torch.int64, device="cuda")
    for _ in range(3):
        plan.run(out)
        fplan.run(fout)
        assert out.tolist() == [1, 1, 0, -1, 0] and fout.tolist() == [100, 0, -1, 100, -1]
        out.fill_(3)
        fout.fill_(3)
    assert ss.search_batched(hay, None, needles, None, position=pos, hay_ranges=(hb, he), needle_ranges=(nb, ne)).tolist() == [1, 1, 0, -1, 0]
    # graph capture: the plan's run is capturable, the unplanned call says why it is not
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        plan.run(out, stream=st.cuda_stream)             # warm-up outside the capture
    st.synchronize()
    with torch.cuda.graph(g, stream=st):
        plan.run(out, stream=torch.cuda.current_stream().cuda_stream)
        with pytest.raises(ss.SlicesliceError) as e:
            ss.search_batched(hay, None, needles, None, hay_ranges=(hb, he), needle_ranges=(nb, ne))
        assert e.value.code == ss.SS_ERR_ARGUMENT and "hipGraph" in str(e.value)
    for rep in range(5):
        hay[100] = 1 if rep % 2 == 0 else 0              # present / absent, decided between replays
        out.fill_(5)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert out.tolist() == [1 if rep % 2 == 0 else 0, 1, 0, -1, 0], rep
    plan.close()
    fplan.close()


def test_plan_runs_of_many_workgroup_problems_are_one_launch_and_replay(ss):
    """a plan run is ONE launch also where problems are scanned by several workgroups - no publish kernel behind the
    scan.  The run's workgroups agree on a parity from the launch's dispatch identity (batched_kernels.hpp, PlanCtl): consecutive runs
    use the two state words of a problem in turn, the slice-0 workgroup re-arms the other one and initialises the caller's output,
    finders write the output behind the state word.  Answers - flags and leftmost offsets - through eager runs on two streams (two
    hardware queues), hipGraph replays and back again, with the needles put in and taken out between runs; both layouts (contiguous
    runs / round robin)."""
    rng = np.random.default_rng(5)
    for count, each in ((96, 1 << 20), (24, 8 << 20)):
        hay = torch.empty(count * each, dtype=torch.uint8, device="cuda")
        ss.fill_random_device(hay, 0xB0B + count)
        nd = np.frombuffer(bytes(ss.fill_random_host(16 * count, 0xD1CE + count).tobytes()), dtype=np.uint8).copy()
        nd[8::16] = 0xFF                                     # absent until planted: 0xFF never occurs in the haystack
        needles = torch.from_numpy(nd).cuda()
        hoff = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
        noff = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
        where = [int(rng.integers(0, each - 16)) if i % 3 else (0 if i % 2 else each - 16) for i in range(count)]
        idx = (torch.arange(count, device="cuda", dtype=torch.int64) * each + torch.tensor(where, device="cuda", dtype=torch.int64))[:, None] + \
            torch.arange(16, device="cuda", dtype=torch.int64)[None, :]
        saved = hay[idx.reshape(-1)].clone()
        plans = {False: ss.BatchPlan(hay, hoff, needles, noff), True: ss.BatchPlan(hay, hoff, needles, noff, find=True)}
        outs = {False: torch.full((count,), 7, dtype=torch.int32, device="cuda"), True: torch.full((count,), 7, dtype=torch.int64, device="cuda")}
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        graphs = {}
        for find, plan in plans.items():
            with torch.cuda.stream(streams[0]):
                plan.run(outs[find], stream=streams[0].cuda_stream)
            streams[0].synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[0]):
                plan.run(outs[find], stream=torch.cuda.current_stream().cuda_stream)
                plan.run(outs[find], stream=torch.cuda.current_stream().cuda_stream)       # two runs per replay: the parity flips inside the graph too
            graphs[find] = g
        present = np.zeros(count, dtype=bool)
        for rnd in range(10):
            flip = rng.random(count) < (0.5 if rnd else 1.0)    # first round: every needle goes in
            present ^= flip
            hay[idx.reshape(-1)] = saved
            sel = torch.from_numpy(np.nonzero(present)[0]).cuda()
            if sel.numel():
                hay[idx[sel].reshape(-1)] = needles.reshape(count, 16)[sel].reshape(-1)
            torch.cuda.synchronize()
            want_flag = [1 if p else 0 for p in present]
            want_at = [where[i] if present[i] else -1 for i in range(count)]
            for find, plan in plans.items():
                out = outs[find]
                how = rnd % 4
                out.fill_(5)
                torch.cuda.synchronize()
                if how == 0:
                    plan.run(out)
                elif how == 1:
                    with torch.cuda.stream(streams[1]):
                        plan.run(out, stream=streams[1].cuda_stream)
                    streams[1].synchronize()
                elif how == 2:
                    graphs[find].replay()
                else:
                    with torch.cuda.stream(streams[0]):
                        plan.run(out, stream=streams[0].cuda_stream)
                        plan.run(out, stream=streams[0].cuda_stream)
                    streams[0].synchronize()
                torch.cuda.synchronize()
                assert out.tolist() == (want_at if find else want_flag), (count, each, rnd, how, find)
        for plan in plans.values():
            plan.close()
        del hay


def _non_latin(n_bytes, seed):
    """UTF-8-like text in a non-Latin script (as tests/test_gpu_filter_and_configs.py): every other byte is a lead byte 0xD0 / 0xD1 -
    which the static, corpus-free rarity table takes for rare."""
    rng = np.random.default_rng(seed)
    pairs = n_bytes // 2
    lead = rng.choice(np.array([0xD0, 0xD1], dtype=np.uint8), size=pairs, p=[0.6, 0.4])
    trail = (0x80 + np.minimum(rng.geometric(0.08, size=pairs) - 1, 63)).astype(np.uint8)
    a = np.empty(pairs * 2, dtype=np.uint8)
    a[0::2], a[1::2] = lead, trail
    blanks = rng.integers(0, pairs, size=pairs // 7)
    a[2 * blanks] = 0x20
    a[2 * blanks + 1] = 0x20
    return a


def _plan_sample_model(host, begins, ends):
    """numpy restatement of batch_sample_kernel + class_from_count (batched_kernels.hpp): 1,024 samples of 4 KiB, sample j from
    problem j * count / 1024 (j mod count when there are fewer problems than samples) at a pseudo-random offset; 16 classes =
    15 - min(15, whole bits of total / count), never seen = 0 (the rarest)."""
    count = len(begins)
    hist = np.zeros(256, dtype=np.int64)
    reps = (1024 + count - 1) // count
    for j in range(1024):
        if count < 1024 and j >= count * reps:
            break
        prob = j * count // 1024 if count >= 1024 else j % count
        h0, h1 = begins[prob], ends[prob]
        if h1 <= h0:
            continue
        ln = h1 - h0
        off, take = 0, ln
        if ln > 4096:
            frac = ((j * 2654435761) & 0xFFFFFFFF) >> 8
            off, take = ((ln - 4096) * frac) >> 24, 4096
        elif count < 1024 and j >= count:
            continue
        hist += np.bincount(host[h0 + off:h0 + off + take], minlength=256)
    total = int(hist.sum())
    cls = [0 if c == 0 else 15 - min(15, (total // int(c)).bit_length() - 1) for c in hist]
    return hist, cls


def _plan_pair_model(needle, cls):
    """plan_one's choice for a needle of 3 bytes or more whose `position` nobody chose.  Up to 16 bytes: needle[0] + the two rarest
    (class) of the bytes behind it, the later one among equals.  Longer: the last byte, the rarest of the 15 in front of it as the
    first filter byte, and the rarest other byte of the 15 behind that."""
    n = len(needle)
    position, anchor = n - 1, 0
    if position >= 16:
        bc = 255
        for k in range(15):
            if cls[needle[position - 15 + k]] <= bc:
                bc, anchor = cls[needle[position - 15 + k]], position - 15 + k
    lim = min(n - anchor, 16)
    s2 = position - anchor
    if anchor == 0:
        bc = 255
        for k in range(1, lim):
            if cls[needle[k]] <= bc:
                bc, s2 = cls[needle[k]], k
    p3, bc = s2, 255
    for k in range(1, lim):
        if k != s2 and cls[needle[anchor + k]] <= bc:
            bc, p3 = cls[needle[anchor + k]], k
    return {anchor, anchor + s2, anchor + p3}


def _cold_model(needle, tri, cls):
    """numpy-free restatement of batch_cold_kernel: the schedule (indices relative to the first filter byte: the bytes 16..31 first,
    rarest class first, at most ten; then 1..15; then the far ones left over; fifteen in all, the other two filter bytes left out)
    and the needle's bytes for the in-register compare (needles that end within 16 bytes of the first filter byte)."""
    anchor = min(tri)
    rel = needle[anchor:]
    lim = min(len(rel), 32)
    skip = {t - anchor for t in tri}
    ok = [k for k in range(1, lim) if k not in skip]
    far = sorted((k for k in ok if k >= 16), key=lambda k: cls[rel[k]])
    near = sorted((k for k in ok if k < 16), key=lambda k: cls[rel[k]])
    order = far[:10] + near
    order = (order + far[10:])[:15] if len(order) < 15 else order[:15]
    exact, back, tail = 0, 0, bytes(16)
    if len(rel) <= 16:
        back = min(anchor, 16 - len(rel))
        exact = len(rel) + back
        tail = (needle[anchor - back:] + bytes(16))[:16]
    return order, bytes(rel[k] for k in order), exact, back, tail


def test_batch_plan_filter_bytes_follow_the_haystacks_histogram(ss):
    """Row f3 of SURVEY.md 8f for config 5: ss_batch_plan_create samples a byte histogram of the plan's haystacks and the plan kernel
    ranks the needle bytes by it (16 classes) instead of by the static, corpus-free table - on text in a non-Latin script the static
    table filters on the lead bytes that make up half the haystack.  The sampled counts and the chosen bytes equal a numpy
    restatement; a caller's `position` is kept; answers (bool and find) equal Python's on every problem, whichever table chose."""
    count, hay_len = 96, 1 << 20
    host = _non_latin(count * hay_len, 5)
    assert ord("e") not in host
    rng = random.Random(9)
    needles, noff, want = bytearray(), [0], []
    for i in range(count):
        at = 2 * rng.randrange(8, hay_len // 2 - 64) + i * hay_len
        n = rng.choice((6, 12, 16, 40))                    # (40: too long for the in-register compare, bytes 16..31 in the schedule)
        w = bytearray(host[at:at + n].tobytes())
        if i % 3:
            w[rng.randrange(1, n - 1)] = ord("e")          # absent: no 'e' in this haystack - and the static table's most common letter
        needles += w
        noff.append(len(needles))
        want.append(host[i * hay_len:(i + 1) * hay_len].tobytes().find(bytes(w)))
    assert any(w >= 0 for w in want) and any(w < 0 for w in want)
    hay = torch.from_numpy(host).cuda()
    hoff = torch.arange(0, (count + 1) * hay_len, hay_len, dtype=torch.int64, device="cuda")
    nbuf = torch.from_numpy(np.frombuffer(bytes(needles), dtype=np.uint8).copy()).cuda()
    noff_t = torch.tensor(noff, dtype=torch.int64, device="cuda")
    begins = [i * hay_len for i in range(count)]
    _, cls = _plan_sample_model(host, begins, [b + hay_len for b in begins])
    assert cls[ord("e")] == 0 and cls[0xD0] >= 13 and cls[0xD1] >= 12

    with ss.tuning_build():
        for find in (False, True):
            plan = ss.BatchPlan(hay, hoff, nbuf, noff_t, find=find)
            os.environ["SLICESLICE_BATCH_STATIC_CLASSES"] = "1"
            try:
                static = ss.BatchPlan(hay, hoff, nbuf, noff_t, find=find)
            finally:
                del os.environ["SLICESLICE_BATCH_STATIC_CLASSES"]
            with_e = 0
            for i in range(count):
                nd = bytes(needles[noff[i]:noff[i + 1]])
                tri, packed, slices = plan.filter_of(i)
                assert slices >= 1 and set(tri) == _plan_pair_model(nd, cls), (i, nd, tri)
                assert [packed & 0xFF, (packed >> 8) & 0xFF, (packed >> 16) & 0xFF] == [nd[k] for k in tri]
                assert plan.cold_of(i) == _cold_model(nd, tri, cls), (i, nd, tri)        # the ready-made cold part
                stri = static.filter_of(i)[0]
                assert len(nd) - 1 in stri and (stri[0] == 0 or len(nd) > 16), "the static table keeps the reference's pair (0, n-1)"
                if ord("e") in nd and len(nd) <= 16:
                    e_at = nd.index(b"e")
                    with_e += 1
                    assert e_at in tri and e_at not in stri, (i, nd, tri, stri)
            assert with_e > count // 3
            for p in (plan, static):
                out = p.run()
                torch.cuda.synchronize()
                assert out.tolist() == (want if find else [1 if w >= 0 else 0 for w in want])
                p.close()
        # the UNPLANNED calls never wait: the first call names the batch, the second has the sampling launched in front of it, the
        # ones after that choose by its classes - the same classes a plan gets; answers are the same at every stage
        assert ss.batch_classes(hay, hoff)[0] == 0
        for call in range(4):
            assert ss.search_batched(hay, hoff, nbuf, noff_t).tolist() == [1 if w >= 0 else 0 for w in want], call
            torch.cuda.synchronize()
            state, got = ss.batch_classes(hay, hoff)
            assert state == (1 if call == 0 else 3), (call, state)
        assert got == cls
        assert ss.find_batched(hay, hoff, nbuf, noff_t).tolist() == want
        # a caller's position is the caller's: needle[position] stays a first-phase byte, the histogram ranks the others
        pos = torch.tensor([(noff[i + 1] - noff[i]) // 2 for i in range(count)], dtype=torch.int64, device="cuda")
        plan = ss.BatchPlan(hay, hoff, nbuf, noff_t, position=pos)
        for i in range(0, count, 7):
            tri = plan.filter_of(i)[0]
            assert (tri[0] == 0 or noff[i + 1] - noff[i] > 16) and (noff[i + 1] - noff[i]) // 2 in tri
            assert plan.cold_of(i) == _cold_model(bytes(needles[noff[i]:noff[i + 1]]), tri, cls)
        assert plan.run().tolist() == [1 if w >= 0 else 0 for w in want]
        plan.close()

    # aliased ranges (many needles, one text: the reference's i386 loop) are sampled all over the text, not 1,024 times at its start
    text = np.concatenate([np.full(1 << 20, ord("a"), dtype=np.uint8), _non_latin(3 << 20, 6)])
    words = [bytes(text[(1 << 20) + 2 * k * 100:(1 << 20) + 2 * k * 100 + 12]) for k in range(2000)]
    tb = torch.from_numpy(text).cuda()
    hb = torch.zeros(len(words), dtype=torch.int64, device="cuda")
    he = torch.full((len(words),), text.size, dtype=torch.int64, device="cuda")
    wb = torch.from_numpy(np.frombuffer(b"".join(words), dtype=np.uint8).copy()).cuda()
    wo = torch.arange(0, 12 * len(words) + 1, 12, dtype=torch.int64, device="cuda")
    hist, cls = _plan_sample_model(text, [0] * len(words), [text.size] * len(words))
    assert 0 < hist[ord("a")] < hist.sum() // 2 and hist[0xD0] > hist.sum() // 8, "the sample reaches behind the first MiB"
    with ss.tuning_build():
        plan = ss.BatchPlan(tb, None, wb, wo, hay_ranges=(hb, he))
        for i in range(0, len(words), 97):
            assert set(plan.filter_of(i)[0]) == _plan_pair_model(words[i], cls)
        assert plan.run().tolist() == [1] * len(words)
        plan.close()


def test_plans_of_long_problems_change_their_layout_with_what_the_last_run_found(ss):
    """A plan of LONG problems holds two layouts: round robin (side by side through each haystack: the fastest full scan) and eight
    contiguous runs per problem (later runs of a found problem leave at once).  A run tallies the problems it found; a later run
    takes the contiguous runs when at least an eighth were found.  Answers are the same through every switch, for flags
    and offsets, with the needles present, removed and put back."""
    count, each = 160, 2 << 20
    hay = torch.empty(count * each, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, 0xA17)
    at = [(i * 7919) % (each - 16) for i in range(count)]
    idx = (torch.arange(count, device="cuda", dtype=torch.int64) * each + torch.tensor(at, device="cuda", dtype=torch.int64))[:, None] + \
        torch.arange(16, device="cuda", dtype=torch.int64)[None, :]
    needles = hay[idx.reshape(-1)].contiguous()
    hoff = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
    noff = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
    host = hay.cpu().numpy()
    nd = needles.cpu().numpy().tobytes()
    want = [host[i * each:(i + 1) * each].tobytes().find(nd[16 * i:16 * i + 16]) for i in range(count)]
    assert all(w >= 0 for w in want)
    with ss.tuning_build():
        for find in (False, True):
            plan = ss.BatchPlan(hay, hoff, needles, noff, find=find)
            lay = plan.layout()
            assert lay["two"] and lay["slices"][0] > 8 and 1 < lay["slices"][1] <= 8 and not lay["next_is_second"], lay
            expect = want if find else [1] * count
            # (round 6: a run is ONE launch - the first finder of every problem counts it into the run's tally, and the first workgroup
            # of the NEXT run hands that count to the host: what a run found is known to the host one run later)
            for run in range(4):
                assert plan.run().tolist() == expect, (find, run)
                torch.cuda.synchronize()
                lay = plan.layout()
                if run == 0:
                    assert lay["found_last"] == 0 and not lay["next_is_second"], (run, lay)
                else:
                    assert lay["found_last"] == count and lay["next_is_second"], (run, lay)
            # the needles leave their haystacks (one byte of each occurrence changed): the next runs - contiguous runs - find nothing,
            # and once the host has heard of it the plan goes round robin again
            saved = hay[idx[:, 0]].clone()
            hay[idx[:, 0]] ^= 0x5A
            torch.cuda.synchronize()
            h2 = hay.cpu().numpy()
            want2 = [h2[i * each:(i + 1) * each].tobytes().find(nd[16 * i:16 * i + 16]) for i in range(count)]
            expect2 = want2 if find else [1 if w >= 0 else 0 for w in want2]
            assert plan.run().tolist() == expect2
            torch.cuda.synchronize()
            assert plan.layout()["next_is_second"]                  # (the tally at hand is still the run before's)
            assert plan.run().tolist() == expect2
            torch.cuda.synchronize()
            lay = plan.layout()
            assert lay["found_last"] == sum(w >= 0 for w in want2) and not lay["next_is_second"], lay
            assert plan.run().tolist() == expect2
            hay[idx[:, 0]] = saved
            torch.cuda.synchronize()
            assert plan.run().tolist() == expect
            assert plan.run().tolist() == expect
            torch.cuda.synchronize()
            assert plan.layout()["next_is_second"]
            plan.close()
        # a plan of few, very long problems keeps ONE layout (eight runs each would not fill the device)
        few = ss.BatchPlan(hay, (torch.arange(5, dtype=torch.int64) * (40 * each)).cuda(), needles[:64], noff[:5])
        assert not few.layout()["two"]
        few.close()
