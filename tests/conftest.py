import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "recbole-fairrec_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def rccl_world1():
    """A 1-rank RCCL world (every kernel and every collective call of the sharded paths runs; the multi-rank exchange
    schedules are covered over gloo)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    if dist.is_initialized():
        if os.environ.get("FAIRREC_TEST_PG_SYNC", "1") == "1":
            torch.cuda.synchronize()        # nothing of this test is in flight when the communicator goes away
        dist.destroy_process_group()
        if os.environ.get("FAIRREC_TEST_PG_SYNC", "1") == "1":
            torch.cuda.synchronize()
