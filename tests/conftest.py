import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_GEMM_SWITCHES = ("LX_GEMM_BM", "LX_GEMM4", "LX_GEMM4_SK", "LX_GEMM4_FAULT")          # = what read_gemm_env() in csrc/gemm.hip caches


@pytest.fixture(autouse=True)
def _gemm_switches_do_not_leak(request):
    """The library caches its LX_GEMM_* switches (lx_gemm_reload_env re-reads them). A test that changed one and reloaded must not leave
    the cached copy behind for the next test: when the environment differs from what the test started with, or the test reloaded the
    switches under a monkeypatched environment, re-read after the environment is restored."""
    before = {k: os.environ.get(k) for k in _GEMM_SWITCHES}
    yield
    lib_mod = sys.modules.get("loongx_amd._lib")
    if lib_mod is None or not hasattr(lib_mod, "lib"):
        return
    # (monkeypatch's own teardown runs after this fixture's: restore the switches here, then re-read)
    for k, v in before.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    try:
        lib_mod.lib.lx_gemm_reload_env()
    except Exception:
        pass
