import json
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


GEMM_MODES = ("fp32", "bf16x3", "f16x2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line(
        "markers",
        "gemm_modes(*modes): run the test once per GEMM mode (default: fp32, bf16x3 and f16x2 - the library default, "
        "bench.py's comparison mode and bench.py's headline mode) through the `gemm_mode` fixture",
    )


def pytest_generate_tests(metafunc):
    """Tests (or whole modules, through `pytestmark`) marked `gemm_modes` run once per GEMM mode: the parity suite
    has to hold in the mode bench.py is timed in, not only in the library default."""
    marker = metafunc.definition.get_closest_marker("gemm_modes")
    if marker is None or "gemm_mode" not in metafunc.fixturenames:
        return
    for m in metafunc.definition.iter_markers("parametrize"):  # the test names its own modes
        names = m.args[0] if isinstance(m.args[0], (list, tuple)) else [n.strip() for n in m.args[0].split(",")]
        if "gemm_mode" in names:
            return
    metafunc.parametrize("gemm_mode", list(marker.args) or list(GEMM_MODES), indirect=True)


@pytest.fixture
def gemm_mode(request):
    """Selects tf2_gnn_amd's GEMM mode for one test (tfgnn_gemm_set_mode + the f16x2 layer paths), restores it after."""
    mode = getattr(request, "param", None)
    if mode is None:
        yield None
        return
    from tf2_gnn_amd import ops

    prev = ops.set_gemm_mode(mode)
    yield mode
    ops.set_gemm_mode(prev)


@pytest.fixture(scope="session")
def kats():
    return json.loads((ROOT / "tests" / "golden" / "reference_kats.json").read_text())


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    return torch.device("cuda", 0)


@pytest.fixture(autouse=True)
def _rearm_default_gemm_mode(request):
    """GPU tests that do not name a mode run in the library default (f16x2).  The spread guard of that mode is sticky: a test
    that trips it (on purpose or not) must not silently move every later test to bf16x3 - re-arm before each GPU test."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import os

    import torch

    if not torch.cuda.is_available():
        yield
        return
    from tf2_gnn_amd import ops

    ops.set_gemm_mode(os.environ.get("TFGNN_GEMM_MODE") or "f16x2")
    yield
