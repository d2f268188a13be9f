"""The one-process-per-GPU path on real hardware, with as many ranks as the box has GPUs (1 on the
test box): torch.distributed (RCCL) and native RCCL (ss_comm_*) transports of the found flag."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_search_both_transports():
    import torch
    n = max(1, min(torch.cuda.device_count(), 8))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "tests", "_sharded_gpu_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "sharded gpu worker ok" in out.stdout
