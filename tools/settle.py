"""Measurement hygiene shared by the tools: wait until the driver has reclaimed the VRAM of whatever process ran
before.  The reclaim of tens of GiB runs in the background for about a second after a process exits, and a scan
that overlaps it is 3-4 % slower (bench.py carries its own copy of this wait)."""
import os
import time


def wait_for_vram_reclaim(device_index=0, limit_bytes=8 << 30, timeout_s=20.0):
    """Returns the seconds waited (0.0 when the sysfs counter is not available)."""
    import torch
    try:
        pr = torch.cuda.get_device_properties(device_index)
        path = "/sys/bus/pci/devices/%04x:%02x:%02x.0/mem_info_vram_used" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        if not os.path.exists(path):
            return 0.0
    except Exception:
        return 0.0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < timeout_s:
        try:
            if int(open(path).read()) <= limit_bytes:
                break
        except Exception:
            break
        time.sleep(0.05)
    return time.perf_counter() - t0
