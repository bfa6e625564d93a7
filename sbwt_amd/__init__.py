"""sbwt_amd -- MI355X (gfx950) implementation of sBWT's read-mapping hot path.

Thin ctypes binding of the C ABI declared in include/sbwt_amd.h.  The library has no CPU
fallback: every call that needs the GPU fails loudly when libsbwt_amd.so or a gfx950 device
is missing.
"""
from .api import (  # noqa: F401
    SbwtError, Index, Context, Hit, ReadResult, lib, lib_path, pack_reads, words_per_read,
    device_count, build_index, build_index_device, host_tables_free, host_tables_save, HostTables, RF_SUPPRESSED, RF_TRUNCATED,
    compact_expand, map_fastq_multi, MAP_HITS_ONLY, CR_NO_SA, CR_WIDE, CR_UNIQUE, CR_MULTI, COMPACT_DTYPE, WIDE_DTYPE,
)
