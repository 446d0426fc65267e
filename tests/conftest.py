import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _product_library_is_built():
    """libxflow_amd.so, the CLI and the binding demo are build artefacts (git-ignored):
    (re)build them when a source is newer — a no-op on an up-to-date tree; on the GPU box the
    built files travel with the snapshot and are used as they are."""
    import shutil
    from xflow_amd import build
    on_gpu_box = os.path.exists("/dev/kfd")  # file times may not survive the snapshot there
    have_hipcc = os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")
    if not have_hipcc:
        return   # no ROCm here: the oracle-only suites still run, product tests fail on load
    if not on_gpu_box or not os.path.exists(build.LIB):
        build.build(verbose=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def sample_prefixes(tmp_path_factory):
    """train/test prefixes laid out like the reference's data/ dir: the three train
    shards are byte-identical there, so -00001/-00002 are copies."""
    import shutil
    d = tmp_path_factory.mktemp("data")
    for r in range(3):
        shutil.copy(os.path.join(GOLDEN, "small_train-00000"),
                    str(d / ("small_train-%05d" % r)))
    shutil.copy(os.path.join(GOLDEN, "small_test-00000"), str(d / "small_test-00000"))
    return str(d / "small_train"), str(d / "small_test")
