import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # GPU runs: the name of every test as it starts goes to gpurun_out/gpu_suite_faults.log (the GPU box sends that directory back even
    # when the process died: the last name is the test that was running).  The fatal-signal report itself -- pytest's faulthandler: the
    # Python stack -- and the runtime's own last words go to stderr, which pytest.ini no longer lets pytest capture at the fd level
    expr = str(config.getoption("markexpr", "") or "")
    if "gpu" in expr and "not gpu" not in expr:
        out = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(out, exist_ok=True)
            config._anofox_fault_log = open(os.path.join(out, "gpu_suite_faults.log"), "a")
            config._anofox_fault_log.write(f"--- pytest -m '{expr}' pid {os.getpid()}\n")
            config._anofox_fault_log.flush()
        except OSError:
            pass


def pytest_runtest_logstart(nodeid, location):
    # the test that was running when the process died is the last name in the log
    log = getattr(pytest, "_anofox_fault_log_ref", None)
    if log is not None:
        log.write(nodeid + "\n")
        log.flush()


@pytest.hookimpl(trylast=True)
def pytest_sessionstart(session):
    pytest._anofox_fault_log_ref = getattr(session.config, "_anofox_fault_log", None)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hiplib():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    from anofox_forecast_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return lib
