import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """Dump the measured-vs-bar table of the parity tests (tests/helpers.py:margin) next to the other GPU artefacts."""
    from tests import helpers
    if not helpers.MARGINS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "margins.json"), "w") as f:
            json.dump(helpers.MARGINS, f, indent=1, sort_keys=True)
    except OSError:
        pass
