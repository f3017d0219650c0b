"""ka9q-radio_amd -- MI355X-native overlap-save channelizer behind ka9q-radio's filter.h.

The product is the gfx950 library ``libchz_hip.so`` (hand-written HIP kernels + the
C ABI of ``include/chz_engine.h``) and the C drop-in ``libka9q_filter_hip.so``
(``include/ka9q_filter_abi.h``).  This package is the thin host-side mirror used by
the parity tests and ``bench.py``: ctypes bindings (``engine``) and a Python
restatement of the reference's calling convention (``filterapi``) on top of them.

The directory name contains a hyphen, so import it through ``load()`` in
``__graft_entry__.py`` (``importlib``) or put the repo root on ``sys.path`` and use
``importlib.import_module("ka9q-radio_amd")``.
"""
from . import engine, filterapi, sharding  # noqa: F401

__all__ = ["engine", "filterapi", "sharding"]
