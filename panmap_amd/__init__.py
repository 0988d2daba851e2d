"""panmap_amd -- MI355X-native hot path of amkram/panmap (index -> place -> align).

All compute is in libpanmap_amd.so (hand-written HIP for gfx950 behind the C ABI in
include/panmap_amd.h); this package is the thin host-side mirror of the reference's interfaces.
"""
from ._lib import LIB_PATH, PmxError, lib  # noqa: F401  (import fails loudly if the .so is missing)
from .api import (METRICS, Aligner, align_reads_direct, write_bam, records_to_results, REC_DTYPE, Context, Index, Panman, PlacementResult, Placer, ReadSet, TraversalParams,  # noqa: F401
                  concat_reads, extract_read_sequences, format_placement_tsv, place_lite, read_fastq_paired,
                  read_fastx, reverse_complement, FastxReads, read_fastq_paired_native, read_fastx_native)
from .meta import Meta, format_abundance, read_dust  # noqa: F401
from ._lib import MetaParams  # noqa: F401
from .api import reload_options, describe_options  # noqa: F401
from .api import Dist, score_reads_vs_reference, last_error, refine_top_candidates, refine_candidates, refine_placement, format_refined_tsv  # noqa: F401
from ._lib import RefineParams  # noqa: F401
from .synth import simulate_paired_reads, simulate_long_reads  # noqa: F401
