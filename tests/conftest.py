import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")
    config.addinivalue_line("markers", "slow: longer learning-sanity runs")


@pytest.fixture
def tmp_root(tmp_path):
    (tmp_path / "models").mkdir()
    (tmp_path / "logs").mkdir()
    return str(tmp_path)
