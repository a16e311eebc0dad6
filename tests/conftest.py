import os
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore", category=UserWarning)
warnings.filterwarnings("ignore", category=FutureWarning)


def _is_xdist_worker(config):
    return bool(os.environ.get("PYTEST_XDIST_WORKER")) or hasattr(config, "workerinput")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (-m "not gpu") spends its time in the SIMT emulator and the CPU oracle: spread it over
    pytest-xdist worker processes when xdist is installed.  Never inside a worker (xdist re-runs this hook there:
    without the guard every worker would spawn workers of its own), never for the GPU suite (one GPU, one
    process), never when -n / -p no:xdist was given.  MN_TEST_WORKERS=0 disables it."""
    if _is_xdist_worker(config) or os.environ.get("MN_TEST_PARALLEL_PARENT"):
        return None
    workers = int(os.environ.get("MN_TEST_WORKERS", "4"))
    if workers <= 1 or config.getoption("markexpr", "") != "not gpu":
        return None
    if not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None):
        return None
    import subprocess
    # build the emulator library once, before the workers race for it
    subprocess.call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "tests", "emu"), "libmapnet_emu.so"])
    os.environ["MN_TEST_PARALLEL_PARENT"] = "1"  # inherited by the workers: second guard against nesting
    os.environ.setdefault("MAPNET_EMU_THREADS", "3")
    os.environ.setdefault("OMP_NUM_THREADS", "2")
    config.option.numprocesses = workers
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
