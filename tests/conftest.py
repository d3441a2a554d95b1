"""pytest configuration: markers, repo root on sys.path, golden-fixture helpers."""

import sys
import warnings
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    warnings.filterwarnings("ignore", message="Type of strand")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


import pytest  # noqa: E402


@pytest.fixture(params=[8, 16])
def md_lanes(request):
    """Lanes per nucleotide of the oxDNA step launches (md_step_kernel's GL): the integrator takes 16 for small systems and
    8 for grids that fill the chip; tests that ask for this fixture run on both, whatever the size of their system."""
    from mythos_amd import _lib

    _lib.debug_set("md_lanes", request.param)
    yield request.param
    _lib.debug_set("md_lanes", 0)
