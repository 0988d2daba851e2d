"""ctypes binding of libpanmap_amd.so (C ABI: include/panmap_amd.h).

The product path has no CPU fallback: importing this module fails loudly when the HIP
library has not been built (run `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PMX_LIB_PATH: load another build of the same library (A/B runs of two builds on one GPU box)
LIB_PATH = os.environ.get("PMX_LIB_PATH") or os.path.join(_HERE, "libpanmap_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP extension first (make -C panmap_amd/csrc); "
        "panmap_amd has no CPU fallback")

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

PMX_OK = 0
ERR_NAMES = {-1: "ARG", -2: "IO", -3: "FORMAT", -4: "NO_DEVICE", -5: "DEVICE", -6: "CAPACITY", -7: "UNSUPPORTED"}


class PmxError(RuntimeError):
    def __init__(self, code, where):
        msg = lib.pmx_last_error().decode(errors="replace")
        super().__init__(f"{where}: PMX_ERR_{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class RefineParams(C.Structure):
    """pmx_refine_params (defaults: src/main.cpp:186-190)"""
    _fields_ = [("top_pct", C.c_double), ("max_top_n", C.c_int32), ("neighbor_radius", C.c_int32), ("max_neighbor_n", C.c_int32),
                ("reserved", C.c_int32)]

    def __init__(self, top_pct=0.01, max_top_n=150, neighbor_radius=2, max_neighbor_n=150):
        super().__init__(top_pct, max_top_n, neighbor_radius, max_neighbor_n, 0)


class RefineResult(C.Structure):
    """pmx_refine_result"""
    _fields_ = [("ran", C.c_int32), ("n_candidates", C.c_int32), ("score", C.c_int64 * 5), ("node", C.c_uint32 * 5), ("reserved", C.c_uint32)]


REFINE_SCORE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_int64))


class IndexInfo(C.Structure):
    _fields_ = [("k", C.c_int32), ("s", C.c_int32), ("t", C.c_int32), ("l", C.c_int32),
                ("open_syncmer", C.c_int32), ("hpc", C.c_int32), ("flank_mask", C.c_int32),
                ("reserved", C.c_int32), ("n_nodes", C.c_int64), ("n_changes", C.c_int64)]


class PlaceParams(C.Structure):
    _fields_ = [("seed_mask_fraction", C.c_double), ("min_read_support", C.c_int32),
                ("trim_start", C.c_int32), ("trim_end", C.c_int32), ("dedup_reads", C.c_int32),
                ("force_leaf", C.c_int32), ("min_seed_quality", C.c_int32), ("reserved", C.c_int32 * 2)]


class PlaceResult(C.Structure):
    _fields_ = [("best_score", C.c_double * 5), ("best_index", C.c_uint32 * 5), ("n_tied", C.c_int64 * 5),
                ("n_reads", C.c_int64), ("n_unique_seeds", C.c_int64), ("n_kept_seeds", C.c_int64),
                ("total_seed_freq", C.c_int64), ("min_support", C.c_int64),
                ("log_read_magnitude", C.c_double), ("log_containment_den", C.c_double),
                ("weighted_containment_den", C.c_double)]


class ReadAlign(C.Structure):
    _fields_ = [("pos", C.c_int32), ("rs", C.c_int32), ("re", C.c_int32), ("qs", C.c_int32), ("qe", C.c_int32),
                ("mapq", C.c_uint8), ("rev", C.c_uint8), ("proper_frag", C.c_uint8),
                ("n_cigar", C.c_int32), ("cigar", C.POINTER(C.c_uint32)), ("md", C.c_char_p)]


class AlignPairResult(C.Structure):
    _fields_ = [("r1", ReadAlign), ("r2", ReadAlign), ("mapped", C.c_int)]


class AlignStats(C.Structure):
    _fields_ = [("n_items", C.c_int64), ("dp_pairs", C.c_int64), ("dp_calls", C.c_int64), ("dp_cells", C.c_int64),
                ("dp_rounds", C.c_int64), ("wave_tier_items", C.c_int64), ("general_tier_items", C.c_int64),
                ("compact_tier_items", C.c_int64), ("reserved", C.c_int64 * 8)]


class AlnRecord(C.Structure):
    _fields_ = [("rs", C.c_int32), ("re", C.c_int32), ("qs", C.c_int32), ("qe", C.c_int32),
                ("mapq", C.c_uint8), ("rev", C.c_uint8), ("proper_frag", C.c_uint8), ("mapped", C.c_uint8),
                ("n_cigar", C.c_uint16), ("flags", C.c_uint16), ("cigar_off", C.c_uint32), ("score", C.c_int32)]


_vp, _i64, _i32, _cp = C.c_void_p, C.c_int64, C.c_int, C.c_char_p
_PP = C.POINTER(_vp)

# every symbol include/panmap_amd.h declares: (restype, argtypes)
SIGNATURES = {
    "pmx_last_error": (_cp, []),
    "pmx_version": (_cp, []),
    "pmx_panman_open": (_i32, [_cp, _PP]),
    "pmx_panman_close": (None, [_vp]),
    "pmx_panman_num_nodes": (_i64, [_vp]),
    "pmx_panman_num_blocks": (_i64, [_vp]),
    "pmx_panman_num_columns": (_i64, [_vp]),
    "pmx_panman_node_id": (_cp, [_vp, _i64]),
    "pmx_panman_parent": (_i64, [_vp, _i64]),
    "pmx_panman_find_node": (_i64, [_vp, _cp]),
    "pmx_panman_node_genome": (_i64, [_vp, _i64, _vp, _i64]),
    "pmx_panman_test_invert_block": (_i64, [_vp, _i64, _i32]),
    "pmx_index_build": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _PP]),
    "pmx_index_build_ex": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _PP]),
    "pmx_index_from_arrays": (_i32, [C.POINTER(IndexInfo), _vp, _vp, _vp, _vp, _vp, _PP]),
    "pmx_index_close": (None, [_vp]),
    "pmx_index_get_info": (_i32, [_vp, C.POINTER(IndexInfo)]),
    "pmx_index_parents": (_vp, [_vp]),
    "pmx_index_offsets": (_vp, [_vp]),
    "pmx_index_hashes": (_vp, [_vp]),
    "pmx_index_parent_counts": (_vp, [_vp]),
    "pmx_index_child_counts": (_vp, [_vp]),
    "pmx_ctx_create": (_i32, [_i32, _PP]),
    "pmx_ctx_destroy": (None, [_vp]),
    "pmx_ctx_synchronize": (_i32, [_vp]),
    "pmx_ctx_stream": (_vp, [_vp]),
    "pmx_readset_upload": (_i32, [_vp, _vp, _vp, _i64, _PP]),
    "pmx_readset_wrap_device": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _PP]),
    "pmx_readset_set_qualities": (_i32, [_vp, _vp, _vp]),
    "pmx_readset_pack": (_i32, [_vp, _vp]),
    "pmx_readset_order_pairs": (_i32, [_vp, _vp]),
    "pmx_readset_pack_range": (_i32, [_vp, _vp, _i64, _i64]),
    "pmx_readset_free": (None, [_vp, _vp]),
    "pmx_readset_num_reads": (_i64, [_vp]),
    "pmx_place_create": (_i32, [_vp, _vp, _PP]),
    "pmx_place_free": (None, [_vp, _vp]),
    "pmx_place_reset": (_i32, [_vp, _vp]),
    "pmx_place_add_reads": (_i32, [_vp, _vp, _vp, C.POINTER(PlaceParams)]),
    "pmx_place_add_reads_range": (_i32, [_vp, _vp, _vp, _i64, _i64, C.POINTER(PlaceParams)]),
    "pmx_place_histogram_size": (_i64, [_vp, _vp]),
    "pmx_place_histogram_export": (_i32, [_vp, _vp, _vp, _vp, _i64]),
    "pmx_place_histogram_merge": (_i32, [_vp, _vp, _vp, _vp, _i64]),
    "pmx_place_histogram_export_device": (_i32, [_vp, _vp, _vp, _vp, _i64]),
    "pmx_place_histogram_entries": (_i64, [_vp, _vp]),
    "pmx_place_histogram_export_device_unsorted": (_i32, [_vp, _vp, _vp, _vp, _i64]),
    "pmx_place_histogram_merge_device_parts": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i32, _i32]),
    "pmx_place_histogram_merge_device": (_i32, [_vp, _vp, _vp, _vp, _i64]),
    "pmx_align_copy_records_device": (_i32, [_vp, _vp, _vp, _i64]),
    "pmx_readset_rewrap_device": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64]),
    "pmx_index_save": (_i32, [_vp, _cp, _i32, _i32]),
    "pmx_index_load": (_i32, [_cp, _PP]),
    "pmx_index_read_header": (_i32, [_cp, C.POINTER(IndexInfo), C.POINTER(C.c_int)]),
    "pmx_index_node_id": (_cp, [_vp, _i64]),
    "pmx_align_copy_cigars_device": (_i32, [_vp, _vp, _vp, _i64]),
    "pmx_align_get_stats": (_i32, [_vp, _vp, C.POINTER(AlignStats)]),
    "pmx_align_scoring": (_i32, [_vp, C.POINTER(C.c_int32)]),
    "pmx_align_dp_batch": (_i32, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _i32, C.POINTER(C.c_double)]),
    "pmx_place_score": (_i32, [_vp, _vp, C.POINTER(PlaceParams), _i64, C.POINTER(PlaceResult)]),
    "pmx_place_tied": (_i32, [_vp, _i32, _vp, _i64]),
    "pmx_place_node_outputs": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "pmx_place_kept_seeds": (_i64, [_vp, _vp, _vp, _vp, _i64]),
    "pmx_fastx_read_paired": (_i32, [_cp, _cp, _PP]),
    "pmx_fastx_read": (_i32, [_cp, _PP]),
    "pmx_fastx_num_reads": (_i64, [_vp]),
    "pmx_fastx_views": (_i32, [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                               C.POINTER(C.c_void_p)]),
    "pmx_fastx_free": (None, [_vp]),
    "pmx_write_bam": (_i32, [_cp, _cp, _i64, _i32, C.POINTER(_cp), C.POINTER(_cp), C.POINTER(_cp), C.POINTER(C.c_int),
                             C.POINTER(AlignPairResult), C.c_bool]),
    "pmx_align_reads_direct": (None, [_cp, _cp, _i32, C.POINTER(_cp), C.POINTER(_cp), C.POINTER(_cp),
                                      C.POINTER(C.c_int), C.POINTER(AlignPairResult), C.c_bool, _i32]),
    "pmx_aligner_create": (_i32, [_vp, _cp, _i64, _i32, _PP]),
    "pmx_aligner_set_reference": (_i32, [_vp, _vp, _cp, _i64, _i32]),
    "pmx_aligner_index_digest": (_i32, [_vp, _vp, _vp]),
    "pmx_aligner_free": (None, [_vp, _vp]),
    "pmx_align_readset": (_i32, [_vp, _vp, _vp, _i32, _i32]),
    "pmx_align_score_reads": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "pmx_refine_candidates": (_i64, [_vp, _i64, _vp, _vp, _vp, _vp, _i64]),
    "pmx_refine_top_candidates": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "pmx_score_reads_vs_reference": (_i64, [_cp, _i32, _vp, _vp, _i32, C.c_bool]),
    "pmx_align_num_records": (_i64, [_vp]),
    "pmx_align_cigar_words": (_i64, [_vp, _vp]),
    "pmx_align_fetch": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64]),
    "pmx_align_fetch_async": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "pmx_align_device_records": (_vp, [_vp]),
    "pmx_align_device_cigars": (_vp, [_vp]),
    "pmx_last_kernel_ms": (C.c_double, [_vp, _cp]),
    "pmx_options_reload": (None, []),
    "pmx_options_describe": (_i64, [_vp, _i64]),
    "pmx_meta_create": (_i32, [_vp, _vp, _vp, _vp]),
    "pmx_meta_free": (None, [_vp, _vp]),
    "pmx_meta_set_reads": (_i32, [_vp, _vp, _vp, _vp, _i64]),
    "pmx_meta_score": (_i32, [_vp, _vp, _i64, _vp, _i64]),
    "pmx_meta_em": (_i32, [_vp, _vp, _vp]),
    "pmx_meta_num_reads": (_i64, [_vp]),
    "pmx_meta_num_candidates": (_i64, [_vp]),
    "pmx_meta_candidates": (_i32, [_vp, _vp, _i64]),
    "pmx_meta_overlap_coefficients": (_i32, [_vp, _vp, _i64]),
    "pmx_meta_read_info": (_i32, [_vp, _vp, _vp, _i64]),
    "pmx_meta_read_seedmers": (_i32, [_vp, _vp, _vp, _vp, _i64]),
    "pmx_meta_scores": (_i32, [_vp, _vp, _vp, _i64]),
    "pmx_meta_num_haplotypes": (_i64, [_vp]),
    "pmx_meta_haplotype": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i64]),
    "pmx_meta_em_info": (_i32, [_vp, _vp, _vp, _vp]),
    "pmx_meta_set_dust": (_i32, [_vp, C.c_double]),
    "pmx_read_dust": (C.c_double, [_cp, _i64, _i32]),
    "pmx_dist_unique_id": (_i32, [_vp]),
    "pmx_dist_init": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "pmx_dist_free": (None, [_vp]),
    "pmx_dist_rank": (_i32, [_vp]),
    "pmx_dist_world": (_i32, [_vp]),
    "pmx_dist_barrier": (_i32, [_vp]),
    "pmx_dist_sum_i64": (_i32, [_vp, _vp, _i64]),
    "pmx_dist_merge_histograms": (_i32, [_vp, _vp]),
    "pmx_dist_dedup_reads": (_i32, [_vp, _vp, _vp, _vp]),
    "pmx_place_dedup_local": (_i64, [_vp, _vp, _vp, _vp, _vp, _i64]),
    "pmx_place_dedup_drop_seen": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64]),
    "pmx_place_dedup_local_count": (_i64, [_vp, _vp, _vp]),
    "pmx_dist_gather_alignments": (_i32, [_vp, _vp, _i32, _vp, _vp]),
    "pmx_dist_plan_alignments": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "pmx_dist_fetch_shard_async": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "pmx_dist_gathered_records": (_vp, [_vp]),
    "pmx_dist_gathered_cigars": (_vp, [_vp]),
    "pmx_dist_rank_counts": (_i32, [_vp, _vp, _vp]),
    "pmx_dist_fetch_gathered": (_i32, [_vp, _vp, _i64, _vp, _i64]),
    "pmx_dist_fetch_gathered_async": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp]),
}

class MetaParams(C.Structure):
    _fields_ = [("error_rate", C.c_double), ("em_convergence", C.c_double), ("em_delta_threshold", C.c_double), ("prop_threshold", C.c_double),
                ("discard", C.c_double), ("em_max_iterations", C.c_int32), ("em_max_rounds", C.c_int32), ("reserved", C.c_int64 * 2)]

    def __init__(self, error_rate=0.005, em_convergence=1e-5, em_delta_threshold=0.0, prop_threshold=0.005, discard=0.0, em_max_iterations=1000,
                 em_max_rounds=5):
        super().__init__(error_rate, em_convergence, em_delta_threshold, prop_threshold, discard, em_max_iterations, em_max_rounds)


MISSING = []
for _name, (_res, _args) in SIGNATURES.items():
    try:
        _f = getattr(lib, _name)
    except AttributeError:
        MISSING.append(_name)
        continue
    _f.restype = _res
    _f.argtypes = _args


def check(code, where):
    if code != PMX_OK:
        raise PmxError(code, where)
