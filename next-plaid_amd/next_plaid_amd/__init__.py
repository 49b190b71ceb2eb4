"""MI355X-native PLAID search path for next-plaid: Python host mirror of the crate API.

`MmapIndex.load / search / search_batch`, `SearchParameters`, `QueryResult` keep the names,
argument meaning and error behaviour of next-plaid/src/{index,search,error}.rs and forward to
the C ABI in include/nextplaid_hip.h (libnextplaid_hip.so, hand-written HIP for gfx950).
There is no CPU fallback in this package: if the HIP library or a GPU is missing, calls raise.
"""
from .api import (  # noqa: F401
    MmapIndex, SearchParameters, QueryResult, NextPlaidError, IndexLoadError, SearchError,
    ShapeError, CodecError, DeviceUnavailableError, device_count, library_path, rerank_maxsim, probe_index_dir, write_index_dir,
)
