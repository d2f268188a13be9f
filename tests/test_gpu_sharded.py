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


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_kernels_with_several_ranks_on_one_gpu(world):
    """world_size 2 and 3 with the REAL shard kernels: the ranks share cuda:0 and the flag all-reduce runs
    over gloo (RCCL refuses two ranks on one GPU).  Matches are planted across every shard boundary."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", SS_TEST_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29540 + world), os.path.join(ROOT, "tests", "_sharded_gpu_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "sharded gpu worker ok" in out.stdout
