import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Borderline-pixel census of the image parity tests (tests/test_gpu_raster.py::_check_image): how many pixels sat
    within 2e-5 of a hard threshold in the oracle, and how many of those actually differed by more than 1e-5."""
    import sys
    mod = sys.modules.get("test_gpu_raster") or sys.modules.get("tests.test_gpu_raster")
    rows = getattr(mod, "BORDERLINE", None) if mod else None
    if rows:
        terminalreporter.write_sep("-", "borderline pixels (oracle threshold margin < 2e-5): test, count / pixels, of which > 1e-5 off, "
                                        "worst |delta| (each bounded by its borderline splat's contribution)")
        for name, nb, npx, nd, worst in rows:
            terminalreporter.write_line(f"{name}: {nb} / {npx}, {nd} differ, worst {worst:.3e}")
        terminalreporter.write_line(f"total: {sum(r[1] for r in rows)} borderline of {sum(r[2] for r in rows)} pixels, "
                                    f"{sum(r[3] for r in rows)} differ by more than 1e-5, worst {max(r[4] for r in rows):.3e}")
