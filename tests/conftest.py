import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pathfinder.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


# the library honours its PFMI_* test hooks from the environment only in a process started with PFMI_DEBUG_HOOKS=1 (include/pfmi.h:
# pfmi_debug_set); the tests select kernels / the RCCL stand-in through them, also in the subprocesses they spawn
os.environ.setdefault("PFMI_DEBUG_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def pfmi_mod():
    import pfmi
    return pfmi


@pytest.fixture(scope="module")
def eng(pfmi_mod):
    """one engine on GPU 0 per test module"""
    e = pfmi_mod.Engine(0)
    yield e
    e.close()


def pytest_collection_modifyitems(session, config, items):
    """The BASELINE configurations (tests/test_gpu_configs.py) run FIRST: with `-x` a failure in a component file then still leaves the
    configuration-level verdict in the log.  The order inside every file is kept."""
    items.sort(key=lambda it: 0 if os.path.basename(str(it.fspath)) == "test_gpu_configs.py" else 1)


def pytest_sessionfinish(session, exitstatus):
    """parity margins recorded by tests/margins.py -> gpurun_out/parity_margins.json (merged back by gpurun)"""
    try:
        import margins
        margins.dump(os.path.join(ROOT, "gpurun_out", "parity_margins.json"))
    except Exception as ex:  # pragma: no cover
        print(f"parity margins not written: {ex!r}")
