import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _effective_cores() -> int:
    """min(cpu_count, affinity, cgroup quota): the GPU box shows 256 CPUs under
    a 16-CPU quota, and 256 OpenMP threads on 16 CPUs make every oracle call
    several times slower (the trajectory test: 2.5 s instead of 0.6 s per step)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    import torch
    torch.set_num_threads(_effective_cores())


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_runtest_logreport(report):
    """Every failure's traceback and captured output also go to
    gpurun_out/pytest_failures.txt (merged back from the GPU box): in round 5 two
    leases went into a flaky test of which the `-q` log kept only the name."""
    if not report.failed:
        return
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "pytest_failures.txt"), "a") as f:
            f.write(f"=== {report.nodeid} [{report.when}]\n{report.longreprtext[-6000:]}\n")
            for name, text in report.sections:
                f.write(f"--- {name}\n{text[-6000:]}\n")
    except OSError:
        pass


@pytest.fixture(autouse=True)
def _library_env_snapshot_follows_monkeypatch():
    """libucsa_hip.so snapshots its UCSA_* switches once per process; tests that
    flip one (monkeypatch.setenv + ops.env_reload()) must not leak it into the next
    test: re-read after every test, once monkeypatch has restored os.environ."""
    yield
    mod = sys.modules.get("ucsa_neural_rendering_amd._lib")
    if mod is not None and getattr(mod, "_lib", None) is not None:
        mod._lib.ucsa_env_reload()
