import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/libmn_oracle.so), built on demand.  Checker only."""
    from oracle import orc as _orc

    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def mn():
    """The product package (sqlite-muninn_amd) over libmuninn_hip.so."""
    import muninn_amd

    pkg = muninn_amd.pkg
    pkg.build()  # no-op unless libmuninn_hip.so is missing or older than its sources (hipcc cross-compiles gfx950)
    return pkg


@pytest.fixture(scope="session")
def gpu(mn):
    """Fails loudly (no skip, no fallback) when the HIP library or a gfx950 device is missing."""
    mn.lib()
    assert mn.device_count() >= 1, "no gfx950 device visible: GPU tests cannot run"
    return mn


@pytest.fixture
def ext_conn(mn):
    """A connection with the SQLite extension (sqlite-muninn_amd/ext/muninn.so) loaded, as the reference is loaded."""
    import sqlite3
    import subprocess

    ext_dir = os.path.join(ROOT, "sqlite-muninn_amd", "ext")
    mn.build()
    subprocess.run(["make", "-s", "-C", ext_dir], check=True)
    c = sqlite3.connect(":memory:")
    c.enable_load_extension(True)
    c.load_extension(os.path.join(ext_dir, "muninn"))
    yield c
    c.close()
