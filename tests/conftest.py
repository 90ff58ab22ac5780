import os
import sys

import pytest
# torch first: it ships its own HIP runtime (libamdhip64 of ROCm 7.0) and only finds the GPU when that copy is the one the
# process loaded; librtamd.so, built against /opt/rocm (7.2), runs on either.  Loading librtamd.so before torch made
# torch.cuda report "No HIP GPUs are available" in tests that hand torch device buffers to the *_dev entry points.
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def have_gpu():
    from raytracer_amd import api
    return api.lib().rt_device_count() > 0
