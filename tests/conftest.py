import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")          # before the HIP runtime initialises (see _lib.py)
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch CPU convolutions (the oracle) get slower, and much noisier, beyond ~16 threads on the 256-thread GPU hosts
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


# GPU suite order: what pins results to the reference / oracle first (train steps, committed fixtures, workflows), the property and
# schedule-equivalence tests last -- a run cut short by a time limit has still judged parity.  Stable within a class.
_FIRST = ("train_step", "vs_committed", "vs_reference_goldens", "workflow", "vs_oracle", "matches_oracle", "match_oracle", "fwd_bwd", "golden")
_LAST = ("deterministic", "bit_identical", "bit_for_bit", "schedules", "bilinearity", "properties", "invariance", "agree_with", "forced",
         "equal_the", "equals_single_stream", "probe_", "heavy_tailed")


def _gpu_order(item):
    if item.get_closest_marker("gpu") is None:
        return 1
    name = item.name
    if any(k in name for k in _LAST):
        return 2
    return 0 if any(k in name for k in _FIRST) else 1


def pytest_collection_modifyitems(config, items):
    items.sort(key=_gpu_order)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_pkg():
    """The package directory carries the reference's (hyphenated) name -> import through importlib."""
    import importlib
    return importlib.import_module("automatic-sem-image-segmentation_amd")


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()
