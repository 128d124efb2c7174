import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU tier (`-m "not gpu"`) is dominated by the SIMT emulator's tests (every lane a thread: ~45 CPU-minutes): with pytest-xdist installed and
    no `-n` on the command line it runs on six workers (~12 min on the 8 cores of the build container).  The GPU tier stays one process: its tests
    share one device and some of them time things.  DOJO_TEST_WORKERS=0 switches it off, another number sets the count."""
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER"):
        return None                           # (a worker of such a run: it must not start workers of its own)
    mexpr = getattr(config.option, "markexpr", "") or ""
    if "not gpu" not in mexpr or not hasattr(config.option, "numprocesses") or config.option.numprocesses is not None:
        return None
    want = os.environ.get("DOJO_TEST_WORKERS", "6")
    try:
        n = max(0, min(int(want), os.cpu_count() or 1))
    except ValueError:
        n = 0
    if n > 1:
        config.option.numprocesses = n      # (pytest-xdist's own pytest_cmdline_main, which runs after this one, turns it into `--tx popen` x n, --dist load)
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Atlas in the tests: states around the reference's standing pose (dojo_amd.coords, distribution "standing") -- the gates of the Atlas
    # cases were tuned there.  bench.py's default is BASELINE.md section 3's perturbation ("baseline"); the full-batch parity test
    # (test_parity_at_the_other_baseline_batches) runs Atlas on that one as well, with its own stated gates.
    os.environ.setdefault("DOJO_SYNTH_DISTRIBUTION", "standing")
    mexpr = config.getoption("-m") or ""
    if "gpu" in mexpr and "not gpu" not in mexpr:
        os.environ.setdefault("DOJO_POISON_OUTPUTS", "1")     # (dojo_hip.hip, launch(): unwritten Jacobian entries come back as NaN)
        # PyTorch ships its own HIP runtime: in a process that uses both torch tensors and libdojo_hip.so (the
        # BatchedEnvironment tests; bench.py) torch has to bring the GPU up first, or its later initialisation
        # finds no device (INTEGRATION.md "Using the library next to PyTorch").
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass


def _gpu_available():
    try:
        import dojo_amd.api as api
        return api.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not silently pass on a fallback:
    # gpu tests are only *skipped* when they were not explicitly selected.
    if "gpu" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="gpu test (select with -m gpu)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
