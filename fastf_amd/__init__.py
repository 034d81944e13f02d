"""fastf_amd — MI355X-native bam2db UMI-counting engine (drop-in for fastF's bam2db path).

The product is the C-ABI shared library ``fastf_amd/lib/libfastf_amd.so`` (hand-written
HIP kernels for gfx950 + the C host side, see include/fastf_amd.h).  This package is the
thin Python host mirror used by the tests, bench.py and the multi-GPU launcher; torch is
only plumbing here (device memory, streams, torch.distributed over RCCL).
"""
from ._lib import build, lib, lib_path, FastfError  # noqa: F401
from .engine import Engine, Lists, PinnedBatch, pack_records, draw_threshold, mt_draws  # noqa: F401

__all__ = ["build", "lib", "lib_path", "FastfError", "Engine", "Lists", "PinnedBatch", "pack_records",
           "draw_threshold", "mt_draws"]
