import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Small test genomes would get no seed-extension tables (the library sizes them from the index): ask for 2 + 1
    # letters by default so that the GPU parity tests run the table path, as an hg38-scale run does.  Tests of the
    # bisection-only path pass seed_extension=(0, 0) to Index().
    import abismal_amd.api as api
    api.DEFAULT_SEED_EXTENSION = (2, 1)
    # Likewise the window records (abm_index_set_window_records): asked for by default, for reads of up to 172 bases, so
    # that the suite's batches of such reads run the record-fed kernels; tests of the bit-plane filter pass
    # window_records=0 (tests/test_gpu_window_records.py runs the same reads through both).
    api.DEFAULT_WINDOW_RECORDS = 172


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/), built on demand.  Test infrastructure only."""
    from tests import oracle_binding
    return oracle_binding.load(build=True)


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("abismal"))


@pytest.fixture(scope="session")
def trex_index(oracle, workdir):
    """tRex1.idx built by the oracle's indexer from the committed tRex1.fa fixture."""
    fa = os.path.join(ROOT, "tests", "golden", "tRex1.fa")
    out = os.path.join(workdir, "tRex1.idx")
    oracle.index_build(fa, out, threads=4)
    return out
