import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    # the oracle's torch-CPU ops are small: on the 256-core GPU box every one of them paid for a 256-thread fork / join (bench.py's
    # cpu_baseline sweep finds 8-16 threads the fastest there too); spawned ranks cap themselves (distributed.init_from_env)
    if "OMP_NUM_THREADS" not in os.environ:
        import torch
        torch.set_num_threads(min(16, torch.get_num_threads()))


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
