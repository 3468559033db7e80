import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if not hasattr(config, "workerinput"):
        # test-only checkers (oracle, lane-loop emulation) are built once, before any pytest-xdist worker loads them
        import subprocess
        for d, target in ((ROOT / "oracle", ["libmp2oracle.so"]), (ROOT / "tests" / "emu", [])):
            subprocess.run(["make", "-s", "-C", str(d)] + target, check=False)


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


def golden_cases():
    return sorted(p for p in (ROOT / "tests" / "golden").glob("p*.npz"))
