"""sqlite-muninn_amd — MI355X-native implementation of sqlite-muninn's compute hot path.

The product is the C-ABI library ``libmuninn_hip.so`` (include/muninn_hip.h) plus the SQLite
loadable extension built on it.  This Python package is the host-side mirror used by the tests
and bench: ``hnsw.HnswIndex`` has the reference's hnsw_algo.h surface (src/hnsw_algo.h:55-92).

The directory name contains a hyphen (it is the project name); import it through
``muninn_amd.py`` at the repository root, which registers it as ``sqlite_muninn_amd``.
"""
from .build import LIB, build  # noqa: F401
from .hnsw import (BUILD_BATCHED, BUILD_SEQUENTIAL, METRIC, ORDER_SSE, ORDER_WAVE, HnswIndex, MuninnHipError, ShardedIndex,  # noqa: F401
                   device_count, lib, vec_dist_batch, vec_parse_metric)
from . import graph  # noqa: E402,F401
from .graph import LEIDEN_BATCHED, LEIDEN_SEQUENTIAL, N2V_BATCHED, N2V_SEQUENTIAL, Graph, node2vec_train  # noqa: E402,F401


def __getattr__(name):  # torch is only needed for the multi-GPU plumbing: import it lazily
    if name == "parallel":
        import importlib

        return importlib.import_module("sqlite_muninn_amd.parallel")
    if name == "lfr":
        import importlib

        return importlib.import_module("sqlite_muninn_amd.lfr")
    raise AttributeError(name)
