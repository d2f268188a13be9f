import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sliceslice_rs_amd as ss
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))

def events_ms(fn, reps=15):
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end: fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize(); ms.append(e0.elapsed_time(e1))
    st = []
    for _ in range(3):
        e0.record()
        for _ in range(10): fn()
        e1.record(); e1.synchronize(); st.append(e0.elapsed_time(e1) / 10)
    return float(np.median(ms)), float(np.median(st))

def batched(blob, count, each):
    nd = bytearray(ss.fill_random_host(16 * count, 0x5EED0003).tobytes()); nd[8::16] = b"\xff" * count
    nblob = torch.from_numpy(np.frombuffer(bytes(nd), dtype=np.uint8).copy()).cuda()
    ho = (torch.arange(count + 1, dtype=torch.int64) * each).cuda(); no = (torch.arange(count + 1, dtype=torch.int64) * 16).cuda()
    b = blob[:count * each]
    return events_ms(lambda: ss.search_batched(b, ho, nblob, no))

def single(buf, n):
    nd = bytearray(ss.fill_random_host(16, 0x5EED0002).tobytes()); nd[8] = 0xFF
    s = ss.DynamicHipSearcher.new(bytes(nd)); s.set_timing(True)
    h = buf[:n]
    s.search_in(h)
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end: s.search_in(h)
    ms = []
    for _ in range(20):
        s.search_in(h); ms.append(s.last_kernel_ms())
    return float(np.median(ms))

for label, nbytes in (("4GiB allocation", 4 << 30), ("64GiB allocation", 64 << 30), ("4GiB allocation again (64 GiB still held)", 4 << 30)):
    buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); ss.fill_random_device(buf, 0x5EED0001); torch.cuda.synchronize()
    if nbytes > (8 << 30): keep = buf
    k = single(buf, 1 << 30)
    c5 = batched(buf, 4096, 1 << 20); c1 = batched(buf, 1024, 1 << 20)
    print(json.dumps({"where": label, "single_1GiB_gbps": round((1 << 30) / k / 1e6, 1), "config5_call_ms": round(c5[0], 4), "config5_steady_ms": round(c5[1], 4),
                      "config5_gbps_steady": round((4 << 30) / c5[1] / 1e6, 1), "b1024_call_ms": round(c1[0], 4), "b1024_steady_ms": round(c1[1], 4)}), flush=True)
    if nbytes <= (8 << 30): del buf
