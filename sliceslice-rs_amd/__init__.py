"""sliceslice-rs_amd - MI355X-native drop-in for the substring-search hot path of
cloudflare/sliceslice-rs (`DynamicAvx2Searcher::{new, with_position, search_in}`).

The product is the C-ABI shared library built from ``csrc/`` (declared in
``include/sliceslice_hip.h``); this package is the thin Python host-side mirror of
the reference interface used by the tests and by ``bench.py``.  The directory
name contains a hyphen, so import it through the repo-root shim::

    import sliceslice_rs_amd as ss
    s = ss.DynamicHipSearcher.new(b"ipsum")
    s.search_in(device_tensor_or_bytes)
"""
from ._build import build, library_path                      # noqa: F401
from .searcher import (                                      # noqa: F401
    SS_OK, SS_ERR_POSITION, SS_ERR_ARGUMENT, SS_ERR_NO_DEVICE, SS_ERR_HIP, SS_ERR_RCCL, SS_ERR_NOMEM, SS_ERR_PEER,
    DynamicHipSearcher, HipSearcher, MemchrHipSearcher, PositionError, SlicesliceError, ShardedSearcher, NodeSearcher, shard_range,
    SearchService, BatchPlan, tuning_build, service_build, set_autotune, rccl_info, TuningState, tools_lib, selftest_dpp,
    search_batched, find_batched, batch_classes, search_file, byte_histogram, choose_position, choose_filter_pair, choose_filter_triple, choose_filter_for_position, fill_random_device, fill_random_host, read_ceiling_gbps, device_info, lib,
)
