"""bgt_amd -- MI355X-native implementation of the BGT genotype-matrix read path.

The product is the C-ABI shared library bgt_amd/lib/libbgt_hip.so (include/bgt_hip.h): hand-written
gfx950 kernels for PBWT run-length decode, rank-tracking column reconstruction, sample-subset gather,
2-bit genotype packing and per-group AC/AN reduction.  This Python package is a thin ctypes mirror of
that ABI for tests, the benchmark and multi-GPU launch (torch.distributed over RCCL); it contains no
decoder of its own and raises if the HIP library or a device is missing.
"""
from .hip import (bench_lib, HipEncoder, HipFilter, HipPbf, HipReader, build_host_shell, build_library, host_lib, device_count, library_path, last_error, lib,  # noqa: F401
                  shard_ranges, synth_rows, force_kernels, forced_kernels)

__all__ = ["bench_lib", "HipEncoder", "HipFilter", "HipPbf", "HipReader", "build_host_shell", "build_library", "host_lib", "device_count", "library_path", "last_error", "lib", "shard_ranges", "synth_rows", "force_kernels", "forced_kernels"]
