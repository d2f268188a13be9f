#!/usr/bin/env python3
"""bench.py - haystack GB/s of the MI355X substring scan (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher around it (WORLD_SIZE unset): bench.py starts the N ranks itself
(`torch.distributed.run --nproc-per-node N`, rendezvous on 127.0.0.1) and refuses - non-zero exit, nothing on
stdout - when fewer than N GPUs are visible or when --gpus disagrees with an existing WORLD_SIZE: a line that
says n_gpus N always comes from N ranks on N devices.

One *step* = one complete `search_in` of the whole logical haystack for a 16-byte ABSENT needle
(position = 15, the `new` default): every rank scans its range shard (shards overlap by n-1 bytes),
the found flags are combined by ONE all-reduce(MAX) over RCCL, and the boolean is read back to the
host.  The haystack is synthetic (SURVEY.md 8d generator), generated on the device, resident in HBM
before the timed region starts.  The logical haystack has a FIXED total size (default 64 GiB, the size
BASELINE.json's target is quoted on; it fits one 288 GB MI355X), so scaling is "strong".

N > 1 runs BOTH forms of the sharded search in one invocation (`--forms both`, the default): first one rank per GPU
(`ss_search_sharded`: scan + ncclAllReduce + answer word on each rank's stream), then - the other ranks idle on a CPU barrier, their
shards freed - all N GPUs from rank 0's process (`ss_comm_init_all` = ncclCommInitAll, one issue thread and one stream per device:
the form a drop-in `search_in` over a node calls, and the one without a rendezvous to fail).  The better one is the line's `value`,
the other one's figures sit under `config.other_form`; `--forms multi` / `--forms single` (= `--single-process`) run one.  The
single-process form is also what bench.py falls back to - and says so in `config.launcher` / `config.transport_note` - when the
ranks of the multi-process form cannot be brought up.  An N > 1 line explains itself: `roofline.per_rank` (min / median / max of
every rank's kernel times), `step_breakdown` (slowest rank's kernel vs the rest of a step; the cost of a sharded search of a
4 MiB shard - launch + collective + answer word - beside a plain one; the host's issue times in the single-process form) and
`config.workgroups_per_cu_timed` (what the timed launches ran with).

Before the `--warmup` steps every rank spins `search_in` for >= 100 ms of wall time (untimed; `config.prewarm_ms`): at 8 GPUs
a step is ~1.2 ms, so warm-up + timed region together are shorter than the clocks' ramp after an idle gap.

stdout: ONE JSON line (rank 0).  Everything else goes to stderr.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
SEED_HAY = 0x5EED0001
SEED_NEEDLE = 0x5EED0002


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def absent_needle(ss, n):
    """Generator bytes with one 0xFF byte; 0xFF never occurs in the haystack."""
    nd = bytearray(ss.fill_random_host(n, SEED_NEEDLE).tobytes())
    nd[0 if n == 1 else (1 if n == 2 else n // 2)] = 0xFF
    return bytes(nd)


def physical_cores():
    """Physical cores of this host: distinct (physical id, core id) pairs of /proc/cpuinfo."""
    try:
        seen, phys = set(), None
        with open("/proc/cpuinfo") as fh:
            for l in fh:
                if l.startswith("physical id"):
                    phys = l.split(":", 1)[1].strip()
                elif l.startswith("core id"):
                    seen.add((phys, l.split(":", 1)[1].strip()))
        return len(seen) or None
    except Exception:
        return None


def cpu_baseline(needle, sample_bytes):
    """The oracle's AVX2 restatement of the reference path, timed on this host (kind = "port":
    the Rust reference itself cannot be built in this image)."""
    from oracle import oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t0 = time.perf_counter()
    hay = O.fill_random(sample_bytes, SEED_HAY)
    gen_s = time.perf_counter() - t0
    s = O.OracleSearcher(needle)
    assert s.search_in(hay[: 1 << 20]) is False

    def rates(threads, reps, discard):
        """GB/s of `reps` timed scans after `discard` untimed ones (the first call with a new thread count builds the pool and
        pins its threads; the second finds the pages where the first left them): (min, median, max)."""
        got = []
        for k in range(discard + reps):
            t = time.perf_counter()
            r = s.search_in(hay, threads=threads)
            dt = time.perf_counter() - t
            assert r is False
            if k >= discard:
                got.append(sample_bytes / dt / 1e9)
        got.sort()
        return got[0], got[len(got) // 2], got[-1]
    one = rates(1, 3, 1)[1]
    # The reference is single-threaded; the multi-thread figure uses the same range-shard rule as the GPUs, on a
    # LARGER sample (4x, so that creating the threads does not dominate a few-millisecond scan) that the same
    # number of NUMA-confined threads first touched.  More threads is not always faster (memory-bound): report
    # the best thread count, and every count tried.
    sweep = {}
    counts = sorted({t for t in (8, 16, 32, 64, 128, cores) if 1 < t <= cores})
    mt_bytes = sample_bytes * 4 if counts else 0
    if counts:
        del hay
        t0 = time.perf_counter()
        hay = O.fill_random(mt_bytes, SEED_HAY, threads=cores)
        gen_mt_s = time.perf_counter() - t0
        sample_bytes_1t, sample_bytes = sample_bytes, mt_bytes
        spread = {}
        for th in counts:
            lo, med, hi = rates(th, 7, 2)
            sweep[th] = med
            spread[str(th)] = {"min": round(lo, 1), "median": round(med, 1), "max": round(hi, 1)}
        sample_bytes = sample_bytes_1t
    best_th = max(sweep, key=sweep.get) if sweep else 1
    out = {
        "value": round(sweep.get(best_th, one), 2), "unit": "GB/s", "cores": best_th, "kind": "port",
        "threads_used": best_th, "host_physical_cores": physical_cores(), "host_hardware_threads": cores,
        "single_thread_value": round(one, 2), "hardware_threads": cores,
        "by_threads": {str(k): round(v, 2) for k, v in sweep.items()}, "avx2": bool(O.have_avx2()),
        "sample": "the same synthetic haystack in host RAM, same 16-byte absent needle; C/AVX2 restatement of "
                  "DynamicAvx2Searcher (oracle/sliceslice_oracle.c): %d MiB, median of 3 runs (1 discarded) on 1 thread; %d MiB "
                  "(first touched by %d NUMA-confined threads), per thread count in by_threads the MEDIAN of 7 runs after 2 "
                  "discarded ones (range shards, n-1 overlap, a persistent pool of threads confined to the NUMA node of their "
                  "share; by_threads_spread has min / median / max); value = the fastest thread count's median"
                  % (sample_bytes >> 20, mt_bytes >> 20, cores),
    }
    if counts:
        out["host_generate_mt_s"] = round(gen_mt_s, 2)
        out["by_threads_spread"] = spread
        ks = sorted(sweep)
        dips = [k for a, k in zip(ks, ks[1:]) if sweep[k] < 0.97 * sweep[a]]
        try:
            nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
        except OSError:
            nodes = None
        out["numa_nodes"] = nodes
        if dips:
            out["by_threads_note"] = ("medians are not monotone at %s threads: the pages were first touched by %d threads (one share each, "
                                      "%s NUMA node(s)), so a count that does not divide that partition evenly reads part of its range "
                                      "from the other node(s); the figure to compare with is `value`" % (dips, cores, nodes))
    # BASELINE.json configs[0]: data/i386.txt x data/words.txt, shape of bench/benches/i386.rs:246-256
    try:
        gd = os.path.join(ROOT, "tests", "golden", "data")
        i386 = open(os.path.join(gd, "i386.txt"), "rb").read()
        words = [w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w]
        iters = 5
        t = time.perf_counter()
        hits = O.bench_long(i386, words, iters)
        out["i386_long_ms_per_iter"] = round((time.perf_counter() - t) / iters * 1e3, 2)
        out["i386_long_hits_per_iter"] = hits // iters
        out["i386_long_published_ms_i7_6700"] = 35.181
        ws = sorted(words, key=len)
        t = time.perf_counter()
        hits = O.bench_short(ws, 2)
        out["i386_short_ms_per_iter"] = round((time.perf_counter() - t) / 2 * 1e3, 2)
        out["i386_short_hits_per_iter"] = hits // 2
        out["i386_short_published_ms_i7_6700"] = 79.416
        # ... and the third criterion group, search_random_haystack (bench/benches/i386.rs:286-289): the same words in data/haystack
        noise = open(os.path.join(gd, "haystack"), "rb").read()
        O.bench_long(noise, words, 20)
        t = time.perf_counter()
        hits = O.bench_long(noise, words, 2000)
        out["i386_random_ms_per_iter"] = round((time.perf_counter() - t) / 2000 * 1e3, 4)
        out["i386_random_hits_per_iter"] = hits // 2000
    except Exception as e:      # pragma: no cover
        out["i386_long_error"] = repr(e)
    out["host_generate_s"] = round(gen_s, 2)
    try:
        with open("/proc/cpuinfo") as fh:
            models = [l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")]
        out["cpu_model"] = models[0] if models else "?"
    except Exception:
        pass
    return out


def cpu_report(out):
    """The extended CPU-side report (`--cpu-report`, no GPU needed): BASELINE.md section 3 timed on THIS host with
    the C/AVX2 restatement - config 1 long + short loops (beside README's 35.181 / 79.416 ms), the needle-length
    sweep {1,2,4,8,16,32,128} at 1 thread and the 16-byte needle at 1 / 8 / 64 / all threads.  JSON lines."""
    from oracle import oracle as O

    def emit(**kw):
        out.write((json.dumps(kw) + "\n").encode())

    def absent(n):
        nd = bytearray(O.fill_random(n, SEED_NEEDLE).tobytes())
        nd[0 if n == 1 else (1 if n == 2 else n // 2)] = 0xFF
        return bytes(nd)
    cores = len(os.sched_getaffinity(0))
    model = "?"
    with open("/proc/cpuinfo") as fh:
        for l in fh:
            if l.startswith("model name"):
                model = l.split(":", 1)[1].strip()
                break
    emit(cpu_model=model, hardware_threads=cores, avx2=bool(O.have_avx2()), note="C restatement of the reference AVX2 path")
    gd = os.path.join(ROOT, "tests", "golden", "data")
    i386 = open(os.path.join(gd, "i386.txt"), "rb").read()
    words = [w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w]
    O.bench_long(i386, words, 1)
    t = time.perf_counter()
    hits = O.bench_long(i386, words, 10)
    emit(config=1, loop="long (bench/benches/i386.rs:246-256)", threads=1, ms_per_iteration=round((time.perf_counter() - t) * 100, 3),
         hits=hits // 10, readme_ms=35.181)
    ws = sorted(words, key=len)
    t = time.perf_counter()
    hits = O.bench_short(ws, 2)
    emit(config=1, loop="short (bench/benches/i386.rs:118-129)", threads=1, ms_per_iteration=round((time.perf_counter() - t) * 500, 3),
         hits=hits // 2, readme_ms=79.416)
    n_bytes = 1 << 30
    hay = O.fill_random(n_bytes, SEED_HAY)
    for n in (1, 2, 4, 8, 16, 32, 128):
        s = O.OracleSearcher(absent(n))
        best = float("inf")
        for _ in range(3):
            t = time.perf_counter()
            r = s.search_in(hay)
            best = min(best, time.perf_counter() - t)
        assert r is False
        emit(config=3, needle_len=n, threads=1, haystack_bytes=n_bytes, gbps=round(n_bytes / best / 1e9, 2))
    s = O.OracleSearcher(absent(16))
    del hay
    mt_bytes = 4 * n_bytes                      # larger sample, first touched by NUMA-confined threads (see cpu_baseline)
    hay = O.fill_random(mt_bytes, SEED_HAY, threads=cores)
    for th in sorted({1, 8, 16, 32, 64, 128, cores}):
        if th > cores:
            continue
        best = float("inf")
        for _ in range(5 if th > 1 else 2):
            t = time.perf_counter()
            r = s.search_in(hay, threads=th)
            best = min(best, time.perf_counter() - t)
        emit(config=2, needle_len=16, threads=th, haystack_bytes=mt_bytes, gbps=round(mt_bytes / best / 1e9, 2))


def fail(msg):
    """Refuse loudly: message on stderr, NOTHING on stdout, non-zero exit."""
    sys.stderr.write("bench.py: " + msg + "\n")
    sys.stderr.flush()
    raise SystemExit(2)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here.  Returns (instead of exiting) when the
    ranks could not be brought up and printed nothing: the caller then runs the single-process form."""
    import subprocess
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    share = os.environ.get("SS_BENCH_SHARE_GPU") == "1"
    if visible < args.gpus and not (share and visible >= 1):
        fail("--gpus %d but only %d HIP device(s) visible; refusing to run a smaller job under that label" % (args.gpus, visible))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: launching %d ranks: %s" % (args.gpus, " ".join(cmd)))
    env = dict(os.environ, SS_BENCH_LAUNCHER="self", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [l for l in r.stdout.decode("utf-8", "replace").splitlines() if l.startswith("{")]
    if r.returncode == 0 and lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
        raise SystemExit(0)
    if lines or share or visible < args.gpus:
        raise SystemExit(r.returncode or 1)
    return "the %d ranks of the multi-process form could not be brought up (torch.distributed.run exit code %d)" % (args.gpus, r.returncode)


def median_kernel_ms(searcher, hay, reps):
    searcher.set_timing(True)
    res = searcher.search_in(hay)
    t_end = time.perf_counter() + 0.05                     # clocks ramp up over the first milliseconds after an idle gap
    while time.perf_counter() < t_end:
        searcher.search_in(hay)
    ms = []
    for _ in range(reps):
        searcher.search_in(hay)
        ms.append(searcher.last_kernel_ms())
    return res, float(np.median(ms))


def _row(nbytes, ms, **kw):
    return dict(kernel_ms=round(ms, 4), gbps=round(nbytes / ms / 1e6, 1), frac=round(nbytes / ms / 1e6 / HBM_PEAK_GBPS, 4), **kw)


def _events_ms(fn, reps):
    """`fn` bracketed by events on torch's current stream: median of `reps` single calls on an otherwise idle stream (includes
    the host's launch path), and per call when 10 are issued back to back (it overlaps the previous call's kernels)."""
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    steady = []
    for _ in range(3):
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        e1.synchronize()
        steady.append(e0.elapsed_time(e1) / 10)
    return float(np.median(ms)), float(np.median(steady))


def other_configs(ss, shard, reps=20):
    """The rest of SURVEY.md 8(d) on one GPU, driver-visible (untimed extras of the N = 1 run).  Kernel time = hipEvents on the
    launch stream, MEDIAN of `reps` launches after a 50 ms spin:
      2_present  the headline workload's 1 GiB sibling with the needle planted at len-16 (true, and the scan's last bytes count);
      3          1 GiB x needle lengths {1,2,4,8,16,32,128}: absent (every byte scanned) and planted at len-n;
      text       English text (i386.txt tiled to 1 GiB): two phrases through `new`, one through the reference's pair (0, n-1);
      adversarial  an all-'a' haystack against a...ab: every offset passes the first byte;
      5          4096 x 1 MiB haystacks x 4096 distinct 16-byte needles in ONE ss_search_batched call;
      5_shapes   the same call on 1 GiB cut into 1,024 / 256 / 64 / 16,384 / 65,536 problems (and 4 GiB in 65,536 x 64 KiB);
      1_short    the reference's short-haystack loop (bench/benches/i386.rs:118-129): 10,513,405 pairs, ss_search_pairs;
      1_random   the reference's third criterion group (bench/benches/i386.rs:286-289): 4,585 words in data/haystack, one call / one plan run."""
    out = {}
    gib = 1 << 30
    gd = os.path.join(ROOT, "tests", "golden", "data")

    def to_dev(b):
        return torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).cuda()

    if shard.numel() >= gib:
        hay = shard[:gib]
        rows = []
        for n in (1, 2, 4, 8, 16, 32, 128):
            s = ss.DynamicHipSearcher.new(absent_needle(ss, n))
            res, ms = median_kernel_ms(s, hay, reps)
            assert res is False
            row = _row(gib, ms, needle_len=n, filter_bytes=list(s.filter3))
            # present variant: a generator-made (0xFF-free) needle planted at len-n - found only by the scan's last chunk
            pres = ss.fill_random_host(n, 0x5EED0003).tobytes()
            saved = hay[gib - n:].clone()
            hay[gib - n:] = to_dev(pres)
            sp = ss.DynamicHipSearcher.new(pres)
            found, pms = median_kernel_ms(sp, hay, 5)
            at = sp.find(hay)
            hay[gib - n:] = saved
            row["present_at_end"] = {"found": found, "kernel_ms": round(pms, 4), "find": at,
                                     "note": None if at == gib - n else "a needle this short occurs earlier in random bytes: early exit"}
            if n >= 8:
                assert found is True and at == gib - n
            rows.append(row)
        out["3"] = {"workload": "1 GiB synthetic haystack (the first GiB of the headline haystack), needles of {1,2,4,8,16,32,128} bytes, "
                                "ss_search_device: absent (0xFF inside), and present (planted at len-n)", "rows": rows}
        r16 = [r for r in rows if r["needle_len"] == 16][0]
        out["2_present"] = {"workload": "1 GiB, 16-byte needle planted at len-16", **r16["present_at_end"]}

    # English text and the adversarial fill (SURVEY.md 8d "extra value distributions")
    try:
        raw = np.frombuffer(open(os.path.join(gd, "i386.txt"), "rb").read(), dtype=np.uint8)
        text = torch.from_numpy(np.tile(raw, gib // raw.size + 1)[:gib].copy()).cuda()
        rows = []
        for nd, how in ((b"privilege level zero!", "new"), (b"segment descriptor table entries are", "new"),
                        (b"segment descriptor table entries are", "with_position(n-1)"),
                        (b" the quick brown fox ", "set_filter(0, n-1): the reference's pair (on this haystack the census replaces its far byte "
                                                    "by near ones - `bytes_in_force`; verbatim under `autotune_off`)")):
            s = ss.DynamicHipSearcher.with_position(nd, len(nd) - 1) if how.startswith("with_position") else ss.DynamicHipSearcher.new(nd)
            if how.startswith("set_filter"):
                s.set_filter(0, len(nd) - 1)
            for _ in range(12):                             # the handle settles on this haystack: census, trials, schedule order
                s.search_in(text)
            res, ms = median_kernel_ms(s, text, reps)
            st = s.tuning_state(text)
            row = _row(gib, ms, needle=nd.decode("latin1"), how=how, found=res, filter_bytes=list(s.filter3),
                       bytes_in_force=st["in_force"], candidate_tiles_of_1024=st["tiles3"], deep_candidates=st["deep_lanes"],
                       schedule_by_census=bool(st["order_measured"]), workgroups_per_cu=s.last_launch()[0],
                       tiles_per_workgroup=max(1, round(text.numel() / 16384 / max(1, s.last_launch()[1]))), kernel_mode=st["kernel_mode"])
            # the same searcher with launch tuning OFF: the static bytes (a caller's pair verbatim, in the cross-lane kernels where it is
            # 16 or more apart), the static schedule, the needle-byte guess for workgroups per CU
            was = ss.set_autotune(False)
            try:
                s0 = ss.DynamicHipSearcher.with_position(nd, len(nd) - 1) if how.startswith("with_position") else ss.DynamicHipSearcher.new(nd)
                if how.startswith("set_filter"):
                    s0.set_filter(0, len(nd) - 1)
                res0, ms0 = median_kernel_ms(s0, text, reps)
                row["autotune_off"] = _row(gib, ms0, found=res0, workgroups_per_cu=s0.last_launch()[0])
            finally:
                ss.set_autotune(was)
            rows.append(row)
        out["text"] = {"workload": "i386.txt (857,425 B of English) tiled to 1 GiB, absent phrases; kernel time by hipEvents once the handle "
                                   "has settled on the haystack (ss_searcher_tuning_state), and the same searcher with ss_set_autotune(0)",
                       "rows": rows}
        del text
        fill = torch.full((gib,), 0x61, dtype=torch.uint8, device="cuda")
        rows = []
        for nd, pos in ((b"a" * 15 + b"b", None), (b"a" * 15 + b"b", 0)):
            s = ss.DynamicHipSearcher(nd, pos)
            res, ms = median_kernel_ms(s, fill, reps)
            rows.append(_row(gib, ms, needle="a" * 15 + "b", how="new" if pos is None else "with_position(0)", found=res,
                             filter_bytes=list(s.filter3)))
        out["adversarial"] = {"workload": "1 GiB of 'a', needle a...ab: every offset passes the first filter byte", "rows": rows}
        del fill
    except Exception as e:      # pragma: no cover
        out["text_error"] = repr(e)

    # row f3 made automatic: a haystack whose most frequent bytes LOOK rare to the static ranking (UTF-8-like text in a non-Latin
    # script: lead bytes 0xD0 / 0xD1, trail bytes 0x80..0xBF, blanks) - `new` (filter bytes re-chosen from the haystack's own
    # histogram, ss_census.hip) next to the same needles with the static triple pinned
    try:
        g = torch.Generator(device="cuda")
        g.manual_seed(7)
        pairs = gib // 2
        lead = (0xD0 + (torch.rand(pairs, device="cuda", generator=g) < 0.4).to(torch.uint8))
        trail = (0x80 + torch.clamp((torch.log(torch.rand(pairs, device="cuda", generator=g).clamp_min(1e-9)) / np.log(1 - 0.08)).floor(), 0, 63)).to(torch.uint8)
        nl = torch.stack((lead, trail), dim=1).reshape(-1).contiguous()
        blanks = torch.randint(0, pairs, (pairs // 7,), device="cuda", generator=g)
        nl[2 * blanks] = 0x20
        nl[2 * blanks + 1] = 0x20
        del lead, trail, blanks
        rows = []
        for k, ln in enumerate((12, 16, 32)):
            w = bytearray(nl[2 * (1000 + 77 * k):2 * (1000 + 77 * k) + ln].cpu().numpy().tobytes())
            w[ln // 2 | 1] = w[1] = 0xBF                       # the rarest trail byte, twice: the word is absent
            auto = ss.DynamicHipSearcher.new(bytes(w))
            pinned = ss.DynamicHipSearcher.new(bytes(w))
            pinned.set_filter(*pinned.filter3)                  # the same static triple, the histogram-driven choice switched off
            for s_ in (auto, pinned):
                for _ in range(4):
                    s_.search_in(nl)
            ra, ma = median_kernel_ms(auto, nl, reps)
            rp, mp = median_kernel_ms(pinned, nl, reps)
            rows.append({"needle_len": ln, "found": ra, "static_filter_bytes": list(pinned.filter3),
                         "automatic": _row(gib, ma, workgroups_per_cu=auto.last_launch()[0]),
                         "static_triple_pinned": _row(gib, mp, workgroups_per_cu=pinned.last_launch()[0])})
            assert ra is rp
        out["text_non_latin"] = {"workload": "1 GiB of UTF-8-like text in a non-Latin script (every other byte a 0xD0 / 0xD1 lead byte), absent words of "
                                             "its alphabet: `new` - filter bytes re-chosen per haystack from a sampled byte histogram - against "
                                             "the static, corpus-free triple pinned with ss_searcher_set_filter3", "rows": rows}
        # ... and the same text as config 5 cuts it: 1,024 haystacks of 1 MiB, an absent 16-byte word of the alphabet each - the call and the
        # plan choose their filter bytes by a byte histogram of the haystacks that the library samples itself (ss_batched.hip)
        cnt, each = 1024, gib // 1024
        at = torch.arange(cnt, device="cuda", dtype=torch.int64) * each + 2 * 1000
        wd = nl[(at[:, None] + torch.arange(16, device="cuda", dtype=torch.int64)[None, :]).reshape(-1)].reshape(cnt, 16).clone()
        wd[:, 1] = 0xBF
        wd[:, 9] = 0xBF
        nblob = wd.reshape(-1).contiguous()
        hay_off = (torch.arange(cnt + 1, dtype=torch.int64) * each).cuda()
        nd_off = (torch.arange(cnt + 1, dtype=torch.int64) * 16).cuda()
        for _ in range(4):
            found = ss.search_batched(nl, hay_off, nblob, nd_off)
        med, steady = _events_ms(lambda: ss.search_batched(nl, hay_off, nblob, nd_off), 15)
        plan = ss.BatchPlan(nl, hay_off, nblob, nd_off)
        flags = torch.empty(cnt, dtype=torch.int32, device="cuda")
        pmed, psteady = _events_ms(lambda: plan.run(flags), 15)
        assert torch.equal(flags, found)
        plan.close()
        out["5_text_non_latin"] = {"workload": "the same text as 1,024 haystacks of 1 MiB, an absent 16-byte word of its alphabet each: one "
                                               "ss_search_batched call / one run of a plan (filter bytes chosen by the haystacks' own sampled "
                                               "byte histogram; the static classes filter this text at 2.7-3.5 TB/s: profiles/r05/batch_triple_probe.jsonl)",
                                   "problems": cnt, "found": int(found.sum().item()), "call_ms": round(med, 4), "frac": round(gib / med / 1e6 / HBM_PEAK_GBPS, 4),
                                   "plan_run_ms": round(pmed, 4), "plan_gbps": round(gib / pmed / 1e6, 1), "plan_frac": round(gib / pmed / 1e6 / HBM_PEAK_GBPS, 4),
                                   "plan_frac_steady": round(gib / psteady / 1e6 / HBM_PEAK_GBPS, 4)}
        del nl
    except Exception as e:      # pragma: no cover
        out["text_non_latin_error"] = repr(e)

    # config 5 and its shapes
    def batched(count, each, total_blob):
        blob = total_blob[:count * each]
        nd = bytearray(ss.fill_random_host(16 * count, SEED_NEEDLE + 1).tobytes())
        nd[8::16] = b"\xff" * count                       # absent: 0xFF never occurs in the haystack
        nblob = to_dev(bytes(nd))
        hay_off = (torch.arange(count + 1, dtype=torch.int64) * each).cuda()
        nd_off = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
        found = ss.search_batched(blob, hay_off, nblob, nd_off)
        assert int(found.sum().item()) == 0
        med, steady = _events_ms(lambda: ss.search_batched(blob, hay_off, nblob, nd_off), 15)
        nb = count * each
        # plan once / search many (ss_batch_plan_*): the reference builds its searchers once and times only the searches
        plan = ss.BatchPlan(blob, hay_off, nblob, nd_off)
        flags = torch.empty(count, dtype=torch.int32, device="cuda")
        plan.run(flags)
        assert int(flags.sum().item()) == 0
        pmed, psteady = _events_ms(lambda: plan.run(flags), 15)
        plan.close()
        return {"problems": count, "haystack_each": each, "call_ms": round(med, 4), "gbps": round(nb / med / 1e6, 1),
                "frac": round(nb / med / 1e6 / HBM_PEAK_GBPS, 4), "steady_call_ms": round(steady, 4),
                "gbps_steady": round(nb / steady / 1e6, 1), "frac_steady": round(nb / steady / 1e6 / HBM_PEAK_GBPS, 4),
                "plan_run_ms": round(pmed, 4), "plan_gbps": round(nb / pmed / 1e6, 1), "plan_frac": round(nb / pmed / 1e6 / HBM_PEAK_GBPS, 4),
                "plan_steady_ms": round(psteady, 4), "plan_frac_steady": round(nb / psteady / 1e6 / HBM_PEAK_GBPS, 4)}
    if shard.numel() >= 4 * gib:
        r = batched(4096, 1 << 20, shard)
        out["5"] = {"workload": "4096 x 1 MiB haystacks, 4096 distinct absent 16-byte needles, ONE ss_search_batched call (plan kernel + "
                                "scan grid; events on the launch stream)", "launch_ms": r["call_ms"], **r,
                    "note": "call_ms: one ss_search_batched call on an idle stream, host launch path included; steady_call_ms: per call "
                            "when 10 are issued back to back; plan_*: the same problems through an ss_batch_plan made once "
                            "(ss_batch_plan_run: one scan launch, no plan kernel, no scratch)"}
        # the yardstick of the 1 GiB cuts: ONE call of the single-problem kernel on the same 1 GiB, bracketed the same way (events on
        # an idle stream, host launch path included) - what "a call on 1 GiB" can reach by this measure
        s1 = ss.DynamicHipSearcher.new(absent_needle(ss, 16))
        one_gib = shard[:gib]
        s1.search_in(one_gib)
        ymed, ysteady = _events_ms(lambda: s1.search_in(one_gib), 15)
        out["5_shapes"] = {"workload": "the same call on other cuts of 1 GiB (and one of 4 GiB): the lengths live on the device, the "
                                       "grid is sized from the problem count alone",
                           "single_problem_1gib_call_ms": round(ymed, 4), "single_problem_1gib_frac": round(gib / ymed / 1e6 / HBM_PEAK_GBPS, 4),
                           "single_problem_1gib_frac_steady": round(gib / ysteady / 1e6 / HBM_PEAK_GBPS, 4),
                           "rows": [batched(c, e, shard) for c, e in ((1024, 1 << 20), (256, 4 << 20), (64, 16 << 20), (16384, 64 << 10),
                                                                     (16384, 256 << 10), (65536, 64 << 10))]}
        for r in out["5_shapes"]["rows"]:
            if r["problems"] * r["haystack_each"] == gib:
                r["single_problem_call_frac"] = out["5_shapes"]["single_problem_1gib_frac"]

    # config 1, short-haystack loop: every needle against every word at or after it in length order (tests/i386.rs:46-59)
    try:
        words = sorted([w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w], key=len)
        W = len(words)
        lens = np.array([len(w) for w in words], dtype=np.int64)
        starts = np.zeros(W, dtype=np.int64)
        starts[1:] = np.cumsum(lens)[:-1]
        blob = to_dev(b"".join(words))
        ni = np.repeat(np.arange(W, dtype=np.int64), W - np.arange(W))
        hj = np.concatenate([np.arange(i, W, dtype=np.int64) for i in range(W)])
        nbt, net = torch.from_numpy(starts[ni]).cuda(), torch.from_numpy(starts[ni] + lens[ni]).cuda()
        hbt, het = torch.from_numpy(starts[hj]).cuda(), torch.from_numpy(starts[hj] + lens[hj]).cuda()
        found = ss.search_batched(blob, None, blob, None, hay_ranges=(hbt, het), needle_ranges=(nbt, net), pairs=True)
        hits = int(found.sum().item())
        med, steady = _events_ms(lambda: ss.search_batched(blob, None, blob, None, hay_ranges=(hbt, het), needle_ranges=(nbt, net), pairs=True), 10)
        out["1_short"] = {"workload": "the reference's short-haystack loop (bench/benches/i386.rs:118-129): every word of words.txt in every "
                                      "word at or after it by length, ONE ss_search_pairs launch (a lane per pair)",
                          "pairs": int(ni.size), "hits": hits, "hits_expected": 39105, "launch_ms": round(med, 4),
                          "ns_per_search": round(med * 1e6 / ni.size, 3), "published_ms_i7_6700": 79.416,
                          "note": "the C restatement's time for the same loop on this host: cpu_baseline.i386_short_ms_per_iter"}
        assert hits == 39105
    except Exception as e:      # pragma: no cover
        out["1_short_error"] = repr(e)

    # config 1, the THIRD criterion group of the reference's harness: search_random_haystack (bench/benches/i386.rs:286-289) - the 4,585
    # words in data/haystack (1,000 bytes of noise), one iteration = every word once (:252-256).  As ONE batched call and as one run of a
    # plan (the needles' searchers built once, :246-250), ranges aliasing the same 1,000 bytes.
    try:
        words = [w for w in open(os.path.join(gd, "words.txt"), "rb").read().split(b"\n") if w]
        noise = open(os.path.join(gd, "haystack"), "rb").read()
        W = len(words)
        lens = np.array([len(w) for w in words], dtype=np.int64)
        starts = np.zeros(W, dtype=np.int64)
        starts[1:] = np.cumsum(lens)[:-1]
        nblob, dh = to_dev(b"".join(words)), to_dev(noise)
        hb, he = torch.zeros(W, dtype=torch.int64, device="cuda"), torch.full((W,), len(noise), dtype=torch.int64, device="cuda")
        nbt, net = torch.from_numpy(starts).cuda(), torch.from_numpy(starts + lens).cuda()
        call = lambda: ss.search_batched(dh, None, nblob, None, hay_ranges=(hb, he), needle_ranges=(nbt, net))   # noqa: E731
        hits = int(call().sum().item())
        med, steady = _events_ms(call, 15)
        plan = ss.BatchPlan(dh, None, nblob, None, hay_ranges=(hb, he), needle_ranges=(nbt, net))
        flags = torch.empty(W, dtype=torch.int32, device="cuda")
        pmed, psteady = _events_ms(lambda: plan.run(flags), 15)
        assert int(flags.sum().item()) == hits
        plan.close()
        want = sum(w in noise for w in words)
        out["1_random"] = {"workload": "the reference's third criterion group, search_random_haystack (bench/benches/i386.rs:286-289): the 4,585 "
                                       "words of words.txt in data/haystack (1,000 bytes of noise); one iteration = every word once",
                           "needles": W, "hits": hits, "hits_expected": want, "call_ms": round(med, 4), "steady_call_ms": round(steady, 4),
                           "plan_run_ms": round(pmed, 4), "plan_steady_ms": round(psteady, 4), "ns_per_search": round(pmed * 1e6 / W, 2),
                           "note": "the C restatement's time for the same loop on this host: cpu_baseline.i386_random_ms_per_iter (the "
                                   "reference publishes no number for this group: README.md:38 has long and short only)"}
        assert hits == want == 106
    except Exception as e:      # pragma: no cover
        out["1_random_error"] = repr(e)
    return out


def native_measurements(ss, headline_gib=None):
    """tools/native_bench (C ABI + HIP runtime only, no Python or torch in the process): per-call latencies, the
    config-1 per-needle loop and - when there is room for a second haystack - the headline measurement itself."""
    import subprocess
    out = {}
    try:
        exe = sys.modules["sliceslice_rs_amd._build"].build_native_bench()
        gd = os.path.join(ROOT, "tests", "golden", "data")
        jobs = [("latency_us", [exe, "latency", "1000"]),
                ("1", [exe, "config1", os.path.join(gd, "i386.txt"), os.path.join(gd, "words.txt"), "3"])]
        if headline_gib:
            jobs.append(("headline_native", [exe, "headline", "%g" % headline_gib, "10"]))
        for key, cmd in jobs:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[key] = json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stderr or r.stdout)[-400:], "rc": r.returncode}
    except Exception as e:      # pragma: no cover
        out["error"] = repr(e)
    return out


def measure_traffic(gib=8.0, launches=6):
    """HBM bytes the headline kernel fetches per haystack byte, measured NOW: a child `rocprofv3 --pmc FETCH_SIZE` pass over this
    script's --traffic-child mode (the same searcher on a `gib` GiB haystack - far beyond the 256 MiB Infinity Cache), counter
    collected and corrected as MI355X_MICROARCH.md's HBM section prescribes (KiB units; x2 on gfx950 for wide coalesced
    streaming reads; a pass of its own with --kernel-trace only).  Returns (ratio, note) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="ss_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "r", "--",
               sys.executable, os.path.abspath(__file__), "--traffic-child", "--haystack-gib", "%g" % gib, "--steps", str(launches)]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, "rocprofv3 --pmc FETCH_SIZE pass failed (rc %d): %s" % (r.returncode, (r.stderr or r.stdout)[-300:])
        vals = [float(x["Counter_Value"]) for f in files for x in csv.DictReader(open(f))
                if "scan_kernel" in x["Kernel_Name"] and x["Counter_Name"] == "FETCH_SIZE"]
        if len(vals) < 2:
            return None, "no FETCH_SIZE rows for ss::scan_kernel"
        fetched = 2.0 * float(np.median(vals)) * 1024.0
        return fetched / (gib * (1 << 30)), ("measured in this run: `rocprofv3 --pmc FETCH_SIZE --kernel-trace` over %d launches of the same "
                                            "kernel on a %g GiB haystack (median FETCH_SIZE %.1f KiB, x2 gfx950 correction), scaled to "
                                            "this run's bytes per launch" % (len(vals), gib, float(np.median(vals))))
    except Exception as e:      # pragma: no cover
        return None, "traffic pass failed: %r" % (e,)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def traffic_child(args):
    """--traffic-child: nothing but `steps` launches of the headline kernel (the parent counts their FETCH_SIZE)."""
    import sliceslice_rs_amd as ss
    torch.cuda.set_device(0)
    n = int(args.haystack_gib * (1 << 30))
    hay = torch.empty(n, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(hay, SEED_HAY)
    torch.cuda.synchronize()
    s = ss.DynamicHipSearcher.new(absent_needle(ss, args.needle_len))
    for _ in range(max(2, args.steps)):
        assert s.search_in(hay) is False


def vram_settle(local_rank, share, settle_seconds):
    """Device hygiene (untimed): the driver reclaims the VRAM of a process that has just exited lazily, and a scan that runs while
    another process' tens of GiB are still being reclaimed is 3-4 % slower (ten back-to-back runs: 7.34-7.38 TB/s with 69 GB of
    VRAM in use afterwards, 7.05-7.13 with 138 GB).  Wait (bounded) until the device's memory is free again before allocating."""
    t_wait = time.perf_counter()
    f = None
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        cand = "/sys/bus/pci/devices/%04x:%02x:%02x.0/mem_info_vram_used" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        if os.path.exists(cand):
            f = cand
    except Exception:
        f = None

    def used():
        try:
            return int(open(f).read())
        except Exception:
            return 0
    at_start = used() if f else None
    while f and not share and used() > (8 << 30) and time.perf_counter() - t_wait < settle_seconds:
        time.sleep(0.05)
    return time.perf_counter() - t_wait, at_start


def prewarm(step, seconds=0.1):
    """>= `seconds` of back-to-back searches before the warm-up steps: the clocks ramp up over the first milliseconds after an
    idle gap, which at 8 GPUs (1.2 ms per step) is longer than warm-up and timed region together."""
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds or n < 3:
        step()
        n += 1
    return (time.perf_counter() - t0) * 1e3, n


def headline_config(args, total, n, shard_bytes, world, filt, info, **kw):
    fa, fb, fc = filt
    cfg = {
        "workload": "%.4g GiB synthetic random-byte haystack (0xFF-free), %d-byte absent needle, `new` (API position %d; "
                    "device filter bytes %d, %d and %d); range-sharded over %d GPU(s) with %d B overlap, one "
                    "all-reduce(MAX) of the found flag" % (total / (1 << 30), n, n - 1, fa, fb, fc, world, n - 1),
        "haystack_bytes": total, "shard_bytes": shard_bytes, "needle_len": n, "filter_bytes": [fa, fb, fc],
        "variant": args.variant, "device": info["name"], "compute_units": info["compute_units"],
        "devices_visible": torch.cuda.device_count(),
    }
    cfg.update(kw)
    return cfg


def rank_stats(per_rank_ms):
    """[{rank, kernel_ms_min / median / max}] from every rank's (device's) list of timed kernel durations."""
    return [{"rank": r, "kernel_ms_min": round(float(np.min(v)), 4), "kernel_ms_median": round(float(np.median(v)), 4),
             "kernel_ms_max": round(float(np.max(v)), 4), "launches": len(v)} for r, v in enumerate(per_rank_ms)]


def step_breakdown(ms_per_step, per_rank_ms, **kw):
    """Where a step's time went: the slowest rank's kernel (median over the timed steps) and everything else - launch, skew between
    the ranks, the collective, the answer word's way to the host."""
    slowest = max(float(np.median(v)) for v in per_rank_ms)
    out = {"kernel_ms_slowest_rank": round(slowest, 4), "outside_kernel_ms": round(ms_per_step - slowest, 4)}
    out.update(kw)
    return out


def roofline_block(shard_bytes, kernel_ms, value, world, traffic_ratio, traffic_source):
    k_med, k_mean = float(np.median(kernel_ms)), float(np.mean(kernel_ms))
    achieved = shard_bytes / (k_med * 1e-3) / 1e9
    return {
        "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "traffic": None if traffic_ratio is None else traffic_ratio * shard_bytes, "traffic_source": traffic_source,
        "traffic_per_algorithmic_byte": None if traffic_ratio is None else round(traffic_ratio, 5),
        "kernel": "ss::scan_kernel", "kernel_ms": round(k_med, 4), "kernel_ms_stat": "median of the timed launches (hipEvents on the launch stream)",
        "kernel_ms_avg": round(k_mean, 4), "kernel_ms_min": round(float(np.min(kernel_ms)), 4), "kernel_launches": len(kernel_ms),
        "algorithmic_bytes_per_launch": shard_bytes,
        "frac_of_whole_job_value": round(value / world / HBM_PEAK_GBPS, 4),
    }


def stored_traffic():
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        pj = json.load(open(tj))
        return pj["hbm_read_bytes_per_haystack_byte"], ("stored ratio, NOT measured in this run: FETCH_SIZE of this kernel from an earlier "
                                                        "`rocprofv3 --pmc FETCH_SIZE` pass (profiles/pmc_traffic.json: %s)" % pj.get("source", "?"))
    except Exception:
        return None, None


def configs_summary(cfg):
    """The other configs in <= 600 bytes, appended as the LAST key of the line: a driver that keeps only the tail of stdout still holds
    every fraction the verdict is written from.  Fractions of 8 TB/s by hipEvents; times in ms."""
    def g(d, *path, default=None):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return default
            d = d[k]
        return d
    r3 = lambda x: None if x is None else round(float(x), 3)     # noqa: E731
    out = {}
    rows3 = g(cfg, "3", "rows", default=[])
    if rows3:
        out["3"] = {str(r["needle_len"]): r3(r["frac"]) for r in rows3}
    rt = g(cfg, "text", "rows", default=[])
    if rt:
        out["text"] = [r3(r["frac"]) for r in rt]
    tn = g(cfg, "text_non_latin", "rows", default=[])
    if tn:
        out["text_non_latin"] = [r3(r["automatic"]["frac"]) for r in tn]
    ad = g(cfg, "adversarial", "rows", default=[])
    if ad:
        out["adversarial"] = [r3(r["frac"]) for r in ad]
    if "5" in cfg:
        out["5"] = {"call": r3(g(cfg, "5", "frac")), "plan": r3(g(cfg, "5", "plan_frac")), "plan_steady": r3(g(cfg, "5", "plan_frac_steady"))}
    rs = g(cfg, "5_shapes", "rows", default=[])
    if rs:
        one = [r for r in rs if r["problems"] * r["haystack_each"] == 1 << 30] or rs
        out["5_shapes_1gib"] = {"call_min": r3(min(r["frac"] for r in one)), "plan_min": r3(min(r["plan_frac"] for r in one)),
                                "plan_steady_min": r3(min(r["plan_frac_steady"] for r in one)),
                                "single_problem_call": r3(g(cfg, "5_shapes", "single_problem_1gib_frac"))}
    out["1_long_ms"] = {"call": r3(g(cfg, "1", "batched_ms_per_iteration")), "plan": r3(g(cfg, "1", "planned_ms_per_iteration"))}
    out["1_short_ms"] = r3(g(cfg, "1_short", "launch_ms"))
    out["1_random_ms"] = {"call": r3(g(cfg, "1_random", "call_ms")), "plan": r3(g(cfg, "1_random", "plan_run_ms"))}
    out["autotune"] = "off" if str(g(cfg, "autotune", default="on")).startswith("off") else "on"
    return out


def write_line(real_stdout, out):
    """THE line: the only bytes this run writes to the real stdout.  `configs_summary` - when the run measured the other configs - is
    its LAST key."""
    if isinstance(out, dict) and isinstance(out.get("configs"), dict):
        out.pop("configs_summary", None)
        out["configs_summary"] = configs_summary(out["configs"])
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(out) + "\n").encode())


def librccl_fields(ss, native):
    """config.librccl_path / librccl_version of an N > 1 line: WHICH librccl the native communicator's ncclAllReduce lives in (dladdr)
    and its ncclGetVersion - a process with torch in it holds torch's bundled library next to the system's."""
    if not native:
        return {"librccl_path": None, "librccl_version": None, "librccl_note": "the flag travels through torch.distributed, not the native communicator"}
    try:
        path, ver = ss.rccl_info()
        return {"librccl_path": path, "librccl_version": ver,
                "librccl_env": os.environ.get("SLICESLICE_RCCL_LIB") or None}
    except Exception as e:      # pragma: no cover
        return {"librccl_path": None, "librccl_version": None, "librccl_note": repr(e)}


def tuning_fields(ss, searcher, shard):
    """config.autotune / config.tuning: which way the timed steps ran - launch tuning on or off, and what the
    handle held about the haystack when the timed region ended (ss_searcher_tuning_state: census counts, the bytes in force, whether
    the second level's schedule is the census's)."""
    try:
        st = searcher.tuning_state(shard)
        keep = ("census_state", "census_age", "tiles3", "tiles2", "lanes", "deep_lanes", "triple_state", "trials", "accepted", "settled", "own",
                "in_force", "order_measured", "histogram_state", "workgroups_per_cu", "kernel_mode")
        return {"autotune": "on" if st["autotune"] else "off", "tuning": {k: st[k] for k in keep}}
    except Exception as e:      # pragma: no cover
        return {"autotune": None, "tuning_note": repr(e)}


def wg_histogram(seen):
    from collections import Counter
    return {str(k): v for k, v in sorted(Counter(seen).items())}


def run_single_process(args, why=None, share=False):
    """All `--gpus` devices from THIS process: ss_comm_init_all (ncclCommInitAll), one shard, one stream and one issue thread per
    device (ss_search_sharded_all).  Returns the line (a dict), `config.launcher` = "single-process"."""
    import sliceslice_rs_amd as ss
    ss.lib()
    G = args.gpus
    if torch.cuda.device_count() < G and not share:
        fail("--gpus %d but only %d HIP device(s) visible" % (G, torch.cuda.device_count()))
    devices = [0] * G if share else list(range(G))
    torch.cuda.set_device(0)
    info = ss.device_info()
    n = args.needle_len
    total = int(args.haystack_gib * (1 << 30))
    waited_s, used_at_start = vram_settle(0, share, args.settle_seconds)
    free_b = min(torch.cuda.mem_get_info(g)[0] for g in set(devices))
    if share:
        free_b //= G
    while (total + G - 1) // G + n > 0.92 * free_b and total > (1 << 28):
        total //= 2
    needle = absent_needle(ss, n)
    note = why
    node = ss.NodeSearcher(needle, devices=devices)            # (ncclCommInitAll; raises when RCCL refuses the set)
    transport = "rccl (ncclCommInitAll; one issue thread per device, each with its own ncclAllReduce)"
    rccl_ranks = node.rccl_ranks() if transport.startswith("rccl") else None      # ncclCommCount of every communicator of the set
    if transport.startswith("rccl") and rccl_ranks != G:
        fail("RCCL reports %r ranks for the set, expected %d" % (rccl_ranks, G))
    if os.environ.get("SLICESLICE_SET_THREADS") == "0":
        transport = transport.replace("one issue thread per device, each with its own ncclAllReduce", "issued from one thread, the all-reduces as one group")
    shards = []
    for g in range(G):
        b, e = node.shard_range(total, g)
        with torch.cuda.device(devices[g]):
            t = torch.empty(e - b, dtype=torch.uint8, device="cuda:%d" % devices[g])
            ss.fill_random_device(t, SEED_HAY, b)
            torch.cuda.synchronize()
        shards.append(t)
    inner = node._searcher
    inner.set_variant(args.variant)
    inner.set_grid(args.grid)
    inner.set_timing(True)

    def sync_all():
        for g in set(devices):
            torch.cuda.synchronize(g)
    prewarm_ms, prewarm_steps = prewarm(lambda: node.search_in(shards))
    for _ in range(args.warmup):
        assert node.search_in(shards) is False
    per_dev = [[] for _ in range(G)]
    issue, wgs = [], []
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        found = node.search_in(shards)                         # G chains (scan -> all-reduce -> answer word) -> bool on the host
        for g, ms in enumerate(node.last_kernel_ms()):         # every device's scan (hipEvents on its stream)
            per_dev[g].append(ms)
        issue.append(node.last_issue_us())
        wgs.append(inner.last_launch()[0])
    sync_all()
    elapsed = time.perf_counter() - t0
    assert found is False
    value = total * args.steps / elapsed / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    # what a search costs besides its scan: the same call on 4 MiB shards (launches + collective + answer words)
    small = [t[: 4 << 20] for t in shards]
    for _ in range(20):
        node.search_in(small)
    t1 = time.perf_counter()
    for _ in range(100):
        node.search_in(small)
    small_us = (time.perf_counter() - t1) / 100 * 1e6
    iss = np.median(np.array(issue), axis=0)
    ratio, src = stored_traffic()
    shard_bytes = max(t.numel() for t in shards)
    slowest = int(np.argmax([np.median(v) for v in per_dev]))
    out = {
        "metric": "haystack GB/s scanned (and % HBM roofline), 16-byte needle, 1/2/4/8 MI355X",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": headline_config(args, total, n, shard_bytes, G, inner.filter3, info, ranks=G, rccl_ranks=rccl_ranks,
                                  **librccl_fields(ss, transport.startswith("rccl")),
                                  transport=transport, transport_note=note, launcher="single-process",
                                  ranks_share_one_gpu=bool(share and G > 1), prewarm_ms=round(prewarm_ms, 1), prewarm_steps=prewarm_steps,
                                  waited_for_free_vram_s=round(waited_s, 2), vram_used_at_start=used_at_start,
                                  workgroups_per_cu_timed=wg_histogram(wgs)),
        "roofline": roofline_block(shard_bytes, per_dev[slowest], value, G, ratio, src),
        "step_breakdown": step_breakdown(ms_per_step, per_dev, small_shard_call_us=round(small_us, 1),
                                         issue_us={"scans": round(float(iss[0]), 1), "collective": round(float(iss[1]), 1),
                                                   "answer_words": round(float(iss[2]), 1), "all": round(float(iss[3]), 1),
                                                   "note": "host time of ss_search_sharded_all's issue phase, median over the timed steps: per-device "
                                                           "maxima when every device has its own issue thread, sums when one thread issues everything"},
                                         note="small_shard_call_us: the same call on 4 MiB shards (launches + collective + answer words; ~1 us of scan)"),
    }
    out["roofline"]["per_rank"] = rank_stats(per_dev)
    out["roofline"]["kernel_ms_of"] = "device %d of the set, the slowest by median" % slowest
    node.close()
    del shards, small
    torch.cuda.empty_cache()
    return out


def other_form_summary(o):
    """What the line keeps of the form that was NOT chosen."""
    keep = {k: o[k] for k in ("value", "ms_per_step") if k in o}
    keep.update(launcher=o["config"].get("launcher"), transport=o["config"].get("transport"), rccl_ranks=o["config"].get("rccl_ranks"),
                transport_note=o["config"].get("transport_note"), step_breakdown=o.get("step_breakdown"),
                per_rank=o["roofline"].get("per_rank"), roofline_frac=o["roofline"].get("frac"),
                workgroups_per_cu_timed=o["config"].get("workgroups_per_cu_timed"))
    return keep


def run_multi_process(args, ctx):
    """One rank per GPU (or the plain N = 1 run): every rank scans its shard and - N > 1 - joins ONE all-reduce per search
    (ss_search_sharded, native RCCL).  Rank 0 returns the line (a dict), the other ranks None."""
    import sliceslice_rs_amd as ss
    dist, backend, world, rank, local_rank, share = ctx["dist"], ctx["backend"], ctx["world"], ctx["rank"], ctx["local_rank"], ctx["share"]
    transport = args.transport
    if share and world > 1 and transport == "rccl" and not os.environ.get("SLICESLICE_RCCL_LIB"):
        transport = "torch"                                    # RCCL refuses two ranks on one device (a stand-in named by SLICESLICE_RCCL_LIB does not)
    ss.lib()
    info = ss.device_info()
    dev = "cuda" if backend == "nccl" else "cpu"

    n = args.needle_len
    total = int(args.haystack_gib * (1 << 30))
    waited_s, used_at_start = vram_settle(local_rank, share, args.settle_seconds)
    free_b, total_b = torch.cuda.mem_get_info()
    if share:
        free_b //= world
    if dist is not None:
        # every rank must partition the SAME logical haystack: agree on the smallest free VRAM first
        fb = torch.tensor([free_b], dtype=torch.int64, device=dev)
        dist.all_reduce(fb, op=dist.ReduceOp.MIN)
        free_b = int(fb.item())
    while (total + world - 1) // world + n > 0.92 * free_b and total > (1 << 28):
        total //= 2                                            # a smaller device: say so in config
    begin, end = ss.shard_range(total, n, world, rank)
    shard = torch.empty(end - begin, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(shard, SEED_HAY, begin)
    torch.cuda.synchronize()
    needle = absent_needle(ss, n)

    rccl_ranks = None
    transport_note = None
    if dist is not None:
        try:
            searcher = ss.ShardedSearcher(needle, group=None, backend=transport)
        except ss.SlicesliceError as e:
            # the native communicator could not be built (librccl not loadable, ncclCommInitRank refused or timed out on
            # some rank): ShardedSearcher makes the ranks agree on that before it raises, so every rank is here and all
            # of them fall back to torch.distributed for the 4-byte flag - and say so
            if transport != "rccl":
                raise
            transport_note = "native RCCL transport failed (%s); flag moved by torch.distributed instead" % e
            log("bench.py: " + transport_note)
            transport = "torch"
            searcher = ss.ShardedSearcher(needle, group=None, backend=transport)
        inner = searcher._searcher
        rccl_ranks = searcher.rccl_ranks()                     # ncclCommCount of the native communicator
        if transport == "rccl" and rccl_ranks != world:
            fail("RCCL reports %r ranks, expected %d" % (rccl_ranks, world))
    else:
        searcher = ss.DynamicHipSearcher.new(needle)
        inner = searcher
    inner.set_variant(args.variant)
    inner.set_grid(args.grid)
    inner.set_timing(True)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # every rank makes the same number of (collective) calls: a fixed count first, then rank 0's clock decides for all
    def one():
        assert searcher.search_in(shard) is False
    t_pw = time.perf_counter()
    pw_steps = 0
    while True:
        for _ in range(8):
            one()
        pw_steps += 8
        go_on = 1 if time.perf_counter() - t_pw < 0.1 else 0
        if dist is not None:
            g = torch.tensor([go_on], dtype=torch.int32, device=dev)
            dist.broadcast(g, src=0)
            go_on = int(g.item())
        if not go_on:
            break
    prewarm_ms = (time.perf_counter() - t_pw) * 1e3
    for _ in range(args.warmup):
        one()
    kernel_ms, wgs = [], []
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        found = searcher.search_in(shard)                      # launch -> (all-reduce) -> bool on the host
        kernel_ms.append(inner.last_kernel_ms())               # hipEvents on the launch stream
        wgs.append(inner.last_launch()[0])
    sync_all()
    elapsed = time.perf_counter() - t0
    assert found is False
    per_rank = [kernel_ms]
    breakdown_extra = {}
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's kernel times, and what a sharded search costs besides its scan: the same collective call on a 4 MiB shard
        # (launch + ~1 us of scan + all-reduce + answer word) next to the plain, rank-local call on the same bytes
        per_rank = [None] * world
        dist.all_gather_object(per_rank, [float(x) for x in kernel_ms])
        small = shard[: 4 << 20]
        for _ in range(20):
            searcher.search_in(small)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(100):
            searcher.search_in(small)
        sharded_us = (time.perf_counter() - t1) / 100 * 1e6
        for _ in range(20):
            inner.search_in(small)
        t1 = time.perf_counter()
        for _ in range(100):
            inner.search_in(small)
        plain_us = (time.perf_counter() - t1) / 100 * 1e6
        tt = torch.tensor([sharded_us, plain_us], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        sharded_us, plain_us = float(tt[0].item()), float(tt[1].item())
        breakdown_extra = dict(small_shard_sharded_call_us=round(sharded_us, 1), small_shard_plain_call_us=round(plain_us, 1),
                               collective_only_us=round(sharded_us - plain_us, 1),
                               note="small_shard_*: a search of a 4 MiB shard (about 1 us of scan), slowest rank, mean of 100 calls - through "
                                    "ss_search_sharded (launch + all-reduce + answer word) and through the rank-local ss_search_device; "
                                    "collective_only_us is their difference")

    ceiling = None
    if not args.no_ceiling and rank == 0:
        ceiling = ss.read_ceiling_gbps(shard[: (shard.numel() // 16) * 16], reps=5)

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total * args.steps / elapsed / 1e9
        ratio, src = stored_traffic()
        out = {
            "metric": "haystack GB/s scanned (and % HBM roofline), 16-byte needle, 1/2/4/8 MI355X",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": headline_config(
                args, total, n, shard.numel(), world, inner.filter3, info,
                ranks=dist.get_world_size() if dist is not None else 1, rccl_ranks=rccl_ranks,
                **(librccl_fields(ss, transport == "rccl") if dist is not None else {}),
                transport=(transport if backend == "nccl" or transport == "rccl" else transport + " over " + backend) if dist is not None else "none",
                transport_note=transport_note,
                launcher=os.environ.get("SS_BENCH_LAUNCHER", "external" if "WORLD_SIZE" in os.environ else "none"),
                ranks_share_one_gpu=bool(share and world > 1), prewarm_ms=round(prewarm_ms, 1), prewarm_steps=pw_steps,
                waited_for_free_vram_s=round(waited_s, 2), vram_used_at_start=used_at_start,
                workgroups_per_cu_timed=wg_histogram(wgs), **tuning_fields(ss, inner, shard)),
            "roofline": roofline_block(shard.numel(), kernel_ms, value, world, ratio, src),
        }
        if dist is not None:
            out["roofline"]["per_rank"] = rank_stats(per_rank)
            out["roofline"]["kernel_ms_of"] = "rank 0"
            out["step_breakdown"] = step_breakdown(ms_per_step, per_rank, **breakdown_extra)
        if ceiling is not None:
            out["roofline"]["read_ceiling_gbps"] = round(ceiling, 2)
            # a diagnosis, not a correction: the scan normally runs at 0.975-0.992 of what a plain streaming read of the same
            # buffer reaches in the same process; far below that the buffer sits badly (a previous process's tens of GiB were
            # still being reclaimed when it was allocated: DESIGN.md section 6)
            out["roofline"]["frac_of_read_ceiling"] = round(out["roofline"]["achieved"] / ceiling, 4)
            if out["roofline"]["achieved"] < 0.96 * ceiling:
                out["roofline"]["placement_note"] = ("the scan kernel reached only %.3f of this buffer's plain-read rate (normally 0.975-0.992): "
                                                     "the haystack was allocated while the driver was still reclaiming another process's memory"
                                                     % (out["roofline"]["achieved"] / ceiling))
        if world == 1 and not args.no_configs:
            cfg = other_configs(ss, shard)
            room = torch.cuda.mem_get_info()[0] > total + (8 << 30)          # a second haystack of the same size fits
            cfg.update(native_measurements(ss, total / (1 << 30) if room else None))
            was_on = ss.set_autotune(True)                      # (read the setting: ss_set_autotune returns the previous one)
            ss.set_autotune(was_on)
            cfg["autotune"] = "on" if was_on else ("off (SLICESLICE_AUTOTUNE=0 / ss_set_autotune(0): static filter bytes, static schedule, "
                                                    "needle-byte guess for workgroups per CU, no sampling kernels)")
            cfg["note"] = ("untimed extras of the N = 1 run; the headline fields above are config 2/4's shape.  3, text, adversarial: "
                           "kernel GB/s by hipEvents (median of 20 after a 50 ms spin); 5, 5_shapes, 1_short: whole calls by events; 1, "
                           "latency_us and headline_native (the headline workload once more, in a process without Python or torch): "
                           "tools/native_bench (C ABI only)")
            out["configs"] = cfg
        if world == 1:
            del shard
            torch.cuda.empty_cache()
            if not args.no_traffic:
                live, why = measure_traffic()
                if live is not None:
                    out["roofline"]["traffic"] = live * out["roofline"]["algorithmic_bytes_per_launch"]
                    out["roofline"]["traffic_per_algorithmic_byte"] = round(live, 5)
                    out["roofline"]["traffic_source"] = why
                else:
                    out["roofline"]["traffic_note"] = "live pass not available (%s)" % why
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(needle, args.cpu_sample_mib << 20)
    if dist is not None and hasattr(searcher, "close"):
        dist.barrier()
        searcher.close()                                       # the native communicator goes before the other form starts
    shard = None
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--haystack-gib", type=float, default=64.0, help="TOTAL logical haystack size")
    ap.add_argument("--needle-len", type=int, default=16)
    ap.add_argument("--transport", choices=["torch", "rccl"], default="rccl",
                    help="N > 1, flag all-reduce: native RCCL via the C ABI (ss_search_sharded: scan + ncclAllReduce + "
                         "read-back on one HIP stream; the default) or torch.distributed (RCCL backend)")
    ap.add_argument("--forms", choices=["both", "multi", "single"], default="both",
                    help="N > 1: one rank per GPU (multi), all GPUs from one process (single), or both - the better one is the "
                         "line's value, the other sits under config.other_form (the default)")
    ap.add_argument("--single-process", action="store_true", help="the same as --forms single")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=1024)
    ap.add_argument("--no-ceiling", action="store_true",
                    help="skip the plain streaming-read ceiling (roofline.read_ceiling_gbps; a few launches, untimed)")
    ap.add_argument("--ceiling", action="store_true", help="(default now) kept for compatibility")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs 1/3/5 + latency block (N = 1 only, untimed)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the live `rocprofv3 --pmc FETCH_SIZE` pass (N = 1 only, untimed); roofline.traffic then uses the stored ratio")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--settle-seconds", type=float, default=20.0,
                    help="upper bound on the wait for a previous process' VRAM to be reclaimed before allocating")
    ap.add_argument("--cpu-report", action="store_true",
                    help="print the extended CPU-side report (JSON lines; no GPU needed) and exit")
    args = ap.parse_args()
    if args.gpus < 1:
        fail("--gpus must be >= 1")
    if args.single_process:
        args.forms = "single"
    if args.cpu_report:
        with os.fdopen(os.dup(1), "wb") as out:
            cpu_report(out)
        return
    if args.traffic_child:
        traffic_child(args)
        return
    fallback_why = None
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and args.forms != "single":
        fallback_why = self_launch(args)                       # returns only when the ranks could not be brought up

    # Exactly ONE line may reach stdout.  Libraries (the RCCL banner, for one) print to fd 1, so fd 1 is
    # pointed at stderr for the whole run and the JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    share = os.environ.get("SS_BENCH_SHARE_GPU") == "1"

    if args.forms == "single" or fallback_why:
        if int(os.environ.get("RANK", "0")) != 0:
            return                                             # under a launcher: rank 0 drives every device, the others have nothing to do
        if not torch.cuda.is_available():
            fail("needs a GPU: the scan has no CPU path")
        write_line(real_stdout, run_single_process(args, fallback_why, share))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        fail("WORLD_SIZE=%d but --gpus %d: refusing to label a %d-rank run as %d GPUs" % (world, args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        fail("needs a GPU: the scan has no CPU path")
    # SS_BENCH_SHARE_GPU=1 + SS_BENCH_BACKEND=gloo: run the N > 1 code path with every rank on cuda:0 (a
    # functional check on a one-GPU box; the numbers of such a run mean nothing and the line says so)
    if share:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        fail("%d ranks but only %d HIP device(s) visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    cpu_group = None
    force_dist = os.environ.get("SS_BENCH_FORCE_DIST") == "1"     # exercise the N > 1 code path on one GPU
    backend = "none"
    if world > 1 or force_dist:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SS_BENCH_BACKEND", "nccl")
        try:
            if os.environ.get("SS_BENCH_FAIL_DIST_INIT") == "1":     # test hook: the bootstrap "fails" on every rank
                raise RuntimeError("SS_BENCH_FAIL_DIST_INIT=1")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                        timeout=datetime.timedelta(seconds=300))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
            probe = torch.zeros(1, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(probe)                             # the first collective is where a broken fabric shows
            if world > 1 and args.forms == "both":
                # the ranks that idle while rank 0 runs the single-process form wait on the CPU: a barrier of the RCCL backend
                # would keep a kernel spinning on every device that is being measured
                cpu_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=1800)) if backend == "nccl" else dist.group.WORLD
        except Exception as e:
            # The ranks cannot talk to each other.  Rank 0 still has every device of the node in reach: it runs the
            # single-process form (no rendezvous at all) and says so; the other ranks have nothing left to do.
            log("bench.py: rank %d: torch.distributed bootstrap failed (%r)" % (rank, e))
            if rank != 0 or share:
                raise SystemExit(0 if not share else 1)
            write_line(real_stdout, run_single_process(args, "torch.distributed bootstrap failed on rank 0 (%s); every device driven from rank 0's "
                                                             "process instead" % (repr(e)[:200],)))
            return

    ctx = dict(dist=dist, backend=backend, world=world, rank=rank, local_rank=local_rank, share=share)
    out = run_multi_process(args, ctx)

    if cpu_group is not None:
        # The other form: every device from ONE process, the ranks parked on a CPU barrier with their shards freed.  It runs in a
        # CHILD of rank 0 with a time limit, so that whatever happens to it - a refusal, a crash inside the collective library, a
        # hang - the multi-process line above still gets printed.
        single = None
        if rank == 0:
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--forms", "single", "--gpus", str(args.gpus), "--steps", str(args.steps),
                   "--warmup", str(args.warmup), "--haystack-gib", "%g" % args.haystack_gib, "--needle-len", str(args.needle_len),
                   "--settle-seconds", "%g" % args.settle_seconds, "--variant", str(args.variant), "--grid", str(args.grid)]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                      "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            try:
                r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=float(os.environ.get("SS_BENCH_SINGLE_TIMEOUT", "600")))
                lines = [l for l in r.stdout.decode("utf-8", "replace").splitlines() if l.startswith("{")]
                if r.returncode == 0 and lines:
                    single = json.loads(lines[-1])
                else:
                    out["config"]["other_form"] = {"launcher": "single-process", "error": "exit code %d, no line" % r.returncode}
            except Exception as e:                              # (TimeoutExpired included)
                log("bench.py: the single-process form failed: %r" % (e,))
                out["config"]["other_form"] = {"launcher": "single-process", "error": repr(e)[:300]}
        dist.barrier(group=cpu_group)
        if rank == 0 and single is not None:
            best, other = (single, out) if single["value"] > out["value"] else (out, single)
            best["config"]["other_form"] = other_form_summary(other)
            best["config"]["forms_run"] = ("one rank per GPU, then all GPUs from one process (a child of rank 0, the ranks parked on a CPU "
                                           "barrier); this line is the better of the two")
            out = best
    if rank == 0:
        write_line(real_stdout, out)
    if dist is not None:
        dist.barrier(group=cpu_group) if cpu_group is not None else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
