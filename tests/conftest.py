import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def rules_golden():
    return load_golden("rules.json")


@pytest.fixture(scope="session")
def search_golden():
    return load_golden("search.json")


@pytest.fixture(scope="session")
def net_golden_meta():
    return load_golden("net_seed0.json")


@pytest.fixture(scope="session")
def sided_golden():
    """Searches captured from the reference with agent.current_player set apart from the mover."""
    return load_golden("search_sided.json")
