import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver at round end)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs must not hold the box: every `gpu` test gets a time limit (pytest-timeout, where it is installed;
    the longest test of the suite takes under a minute, the whole suite three).  One run of round 4 sat in
    test_direct_boundary_is_reentrant for 25 minutes until the outer `timeout` ended it."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(300))


@pytest.fixture(scope="session")
def built():
    """Make sure the HIP library and the oracle are built (cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def pmx(built):
    import panmap_amd
    return panmap_amd


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import oracle as orc
    return orc


@pytest.fixture(scope="session")
def sars(pmx):
    return pmx.Panman(os.path.join(GOLDEN, "sars_20000_twilight_dipper.panman"))


@pytest.fixture(scope="session")
def sars_index(pmx, sars):
    return pmx.Index.build(sars, k=19, s=8, t=0, l=3, open_syncmer=False, flank_mask=250)


@pytest.fixture(scope="session")
def isolate_reads(pmx):
    return pmx.extract_read_sequences(os.path.join(GOLDEN, "isolate_R1.fastq.gz"), os.path.join(GOLDEN, "isolate_R2.fastq.gz"))


@pytest.fixture(scope="session")
def ctx(pmx):
    return pmx.Context(0)


@pytest.fixture(autouse=True)
def _library_options_follow_the_environment(monkeypatch, built):
    """The library reads its PMX_* switches ONCE (csrc/device/pmx_options.hpp).  A test that changes one through
    monkeypatch gets the table re-read at that moment, and once more when the test's changes are undone."""
    import panmap_amd
    real_set, real_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        real_set(name, value, *a, **k)
        if name.startswith("PMX_"):
            panmap_amd.reload_options()

    def delenv(name, *a, **k):
        real_del(name, *a, **k)
        if name.startswith("PMX_"):
            panmap_amd.reload_options()
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.undo()
    panmap_amd.reload_options()
