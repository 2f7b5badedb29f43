import os
import sys

import pytest

# Several tests import torch only inside the test that needs it, after the product library has been loaded by an
# earlier one: ask the loader to map torch's copy of the HIP runtime from the start (two runtimes in one process
# do not work; raytracing_amd/_lib.py).
os.environ.setdefault("RT_AMD_SHARE_TORCH_HIP", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
