"""ctypes binding of libgavisunk.so (include/gavisunk.h).  No fallback: if the library is not
built, importing the engine fails with instructions instead of silently computing on the CPU."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GAVISUNK_LIB") or os.path.join(HERE, "libgavisunk.so")  # override: tuning experiments

GS_OK, GS_E_ARG, GS_E_KEYERROR, GS_E_CAPACITY, GS_E_NODEVICE, GS_E_IO, GS_E_NOMEM, GS_E_IRREGULAR = 0, -1, -2, -3, -4, -5, -6, -7
GS_NONE = 0xFFFFFFFF
GS_MISSING = 0xFFFFFFFE


class GsError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"libgavisunk: {msg} (status {status})")
        self.status = status
        self.msg = msg


class GsHit(C.Structure):
    _fields_ = [("read", C.c_uint32), ("pos", C.c_uint32), ("payload", C.c_uint32)]


class GsCounters(C.Structure):
    _fields_ = [("windows", C.c_uint64), ("raw_hits", C.c_uint64), ("emitted", C.c_uint64), ("tiles", C.c_uint64),
                ("launches", C.c_uint64), ("probe_ms", C.c_double), ("total_ms", C.c_double), ("fallback_tiles", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# name -> (restype, argtypes): every symbol include/gavisunk.h declares
P, U32, U64, I = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
SIGNATURES = {
    "gs_last_error": (C.c_char_p, []),
    "gs_version": (C.c_char_p, []),
    "gs_device_count": (I, []),
    "gs_device_synchronize": (I, [I]),
    "gs_device_blocking_sync": (I, [I, I]),
    "gs_selftest_minimizer_rule": (I, [I, I, I, U64, U32, C.POINTER(U64), C.POINTER(U32), C.POINTER(U32)]),
    "gs_sunkdb_load": (I, [C.c_char_p, C.c_char_p, C.POINTER(P)]),
    "gs_sunkdb_free": (None, [P]),
    "gs_sunkdb_k": (I, [P]),
    "gs_sunkdb_n_keys": (U64, [P]),
    "gs_sunkdb_keys": (P, [P]),
    "gs_sunkdb_key_row": (P, [P]),
    "gs_sunkdb_key_loc": (P, [P]),
    "gs_sunkdb_n_rows": (U64, [P]),
    "gs_sunkdb_row_contig": (P, [P]),
    "gs_sunkdb_row_start": (P, [P]),
    "gs_sunkdb_row_group": (P, [P]),
    "gs_sunkdb_row_loc": (P, [P]),
    "gs_sunkdb_n_contigs": (U32, [P]),
    "gs_sunkdb_contig_name": (C.c_char_p, [P, U32]),
    "gs_fastx_open": (I, [C.c_char_p, C.POINTER(P)]),
    "gs_fastx_close": (None, [P]),
    "gs_fastx_next_batch": (I, [P, U64, U32, C.POINTER(P)]),
    "gs_batch_free": (None, [P]),
    "gs_batch_n_reads": (U32, [P]),
    "gs_batch_n_bases": (U64, [P]),
    "gs_batch_words": (P, [P]),
    "gs_batch_n_words": (U64, [P]),
    "gs_batch_word_off": (P, [P]),
    "gs_batch_len": (P, [P]),
    "gs_batch_name": (C.c_char_p, [P, U32]),
    "gs_batch_names": (I, [P, C.POINTER(P), C.POINTER(P), C.POINTER(U64)]),
    "gs_batch_from_ascii": (I, [P, P, U32, C.POINTER(P)]),
    "gs_rowwriter_open": (I, [C.c_char_p, C.POINTER(P)]),
    "gs_rowwriter_rows": (I, [P, P, U64, P, P, P, P, P]),
    "gs_rowwriter_close": (I, [P, C.POINTER(U64)]),
    "gs_rowwriter_rlen": (I, [P, P, U64, C.POINTER(U64)]),
    "gs_rowwriter_rows_device": (I, [P, P, P, P, U32, P, P, P, P, U32, P]),
    "gs_rowwriter_rows_device_cuts": (I, [P, P, P, P, U32, P, P, P, P, U32, P, P, U32, P]),
    "gs_gz_open": (I, [C.c_char_p, I, C.POINTER(P)]),
    "gs_gz_read": (I, [P, U64, P, U64, C.POINTER(P), C.POINTER(U64)]),
    "gs_gz_free_buffer": (None, [P]),
    "gs_gz_at_end": (I, [P]),
    "gs_gz_mode": (C.c_char_p, [P]),
    "gs_gz_pieces": (U64, [P, C.POINTER(U64)]),
    "gs_gz_close": (None, [P]),
    "gs_bgzf_open": (I, [C.c_char_p, I, C.POINTER(P)]),
    "gs_bgzf_next": (I, [P, U64, C.POINTER(P)]),
    "gs_bgzf_at_end": (I, [P]),
    "gs_bgzf_bytes_in": (U64, [P]),
    "gs_bgzf_close": (None, [P]),
    "gs_textstage_names": (I, [P, P, P, U32, P, P]),
    "gs_textstage_copy": (I, [P, P, U64, C.POINTER(U64)]),
    "gs_text_record_cut": (U64, [P, U64]),
    "gs_gz_newlines": (U64, [P]),
    "gs_text_record_cut_counted": (U64, [P, U64, U64]),
    "gs_table_create": (I, [I, I, U64, P, P, P, C.POINTER(P)]),
    "gs_table_create_device": (I, [I, I, U64, P, P, P, C.POINTER(P)]),
    "gs_table_create_from_db": (I, [I, P, C.POINTER(P)]),
    "gs_table_destroy": (None, [P]),
    "gs_table_bytes": (U64, [P]),
    "gs_table_layout": (C.c_char_p, [P]),
    "gs_reads_upload": (I, [I, P, U64, P, P, U32, I, C.POINTER(P)]),
    "gs_reads_upload_ascii": (I, [I, P, P, U32, I, C.POINTER(P)]),
    "gs_reads_upload_ordered": (I, [I, P, U64, P, P, U32, P, U32, I, C.POINTER(P)]),
    "gs_reads_set_seeds": (I, [P, P, P, U32]),
    "gs_tag_segment_carries": (I, [P, P, P]),
    "gs_reads_from_device_ascii": (I, [I, P, P, U32, I, C.POINTER(P)]),
    "gs_reads_from_text": (I, [I, P, U64, I, C.POINTER(P), C.POINTER(P)]),
    "gs_text_stage": (I, [I, P, U64, C.POINTER(P)]),
    "gs_text_stage_free": (None, [P]),
    "gs_reads_from_staged_text": (I, [P, I, C.POINTER(P), C.POINTER(P)]),
    "gs_textindex_free": (None, [P]),
    "gs_textindex_n_reads": (U32, [P]),
    "gs_textindex_name_off": (P, [P]),
    "gs_textindex_name_len": (P, [P]),
    "gs_textindex_len": (P, [P]),
    "gs_reads_destroy": (None, [P]),
    "gs_reads_n_bases": (U64, [P]),
    "gs_reads_n_windows": (U64, [P]),
    "gs_tag": (I, [P, P, U32, P, U64, C.POINTER(U64), C.POINTER(U32), C.POINTER(GsCounters)]),
    "gs_tag_launch": (I, [P, P, U32]),
    "gs_tag_finish": (I, [P, P, P, U64, C.POINTER(U64), C.POINTER(U32), C.POINTER(GsCounters)]),
    "gs_tag_phase1": (I, [P, P, C.POINTER(U32), C.POINTER(U32)]),
    "gs_tag_phase2": (I, [P, P, U32]),
    "gs_tag_phase1_launch": (I, [P, P]),
    "gs_tag_phase1_wait": (I, [P, P, C.POINTER(U32), C.POINTER(U32)]),
    "gs_tag_hits_to_host": (I, [P, P, P, U64, C.POINTER(U64)]),
    "gs_tag_hits_to_device": (I, [P, P, P, U64, C.POINTER(U64)]),
    "gs_tag_launch_after": (I, [P, P, P]),
    "gs_table_counters": (None, [P, C.POINTER(GsCounters)]),
    "gs_reads_copy_words": (I, [P, U64, U64, P]),
    "gs_stream_create": (I, [P, U64, U32, I, U32, C.POINTER(P)]),
    "gs_stream_destroy": (None, [P]),
    "gs_stream_submit": (I, [P, P, U64, P, P, U32]),
    "gs_stream_submit_reads": (I, [P, P]),
    "gs_stream_next": (I, [P, C.POINTER(P), C.POINTER(U64), C.POINTER(U32), C.POINTER(GsCounters)]),
    "gs_stream_release": (None, [P]),
    "gs_stream_in_flight": (I, [P]),
    "gs_stream_carry": (U32, [P]),
    "gs_host_alloc": (I, [U64, C.POINTER(P)]),
    "gs_host_free": (None, [P]),
    "gs_tag_hits_view": (I, [P, P, C.POINTER(P), C.POINTER(U64)]),
    "gs_counters_reset": (None, [P]),
    "gs_comm_unique_id": (I, [P]),
    "gs_comm_init": (I, [I, I, I, P, C.POINTER(P)]),
    "gs_table_clone": (I, [P, C.POINTER(P)]),
    "gs_sidecar_write": (I, [C.c_char_p, P, P, C.c_char_p, C.c_char_p]),
    "gs_sidecar_open": (I, [C.c_char_p, C.c_char_p, C.c_char_p, I, C.POINTER(P), C.POINTER(P)]),
    "gs_comm_allgather_u64": (I, [P, P, U32, P]),
    "gs_comm_allreduce_u32": (I, [P, P, U64]),
    "gs_comm_reduce_u32": (I, [P, P, U64]),
    "gs_comm_gatherv_host": (I, [P, I, P, U64, I, C.POINTER(P), P]),
    "gs_comm_destroy": (None, [P]),
    "gs_table_broadcast": (I, [P, I, P, C.POINTER(P)]),
    "gs_gather_hits": (I, [P, I, P, P, U32, P, U64, C.POINTER(U64)]),
    "gs_gather_device": (I, [P, C.POINTER(P), C.POINTER(U64)]),
    "gs_gather_view": (I, [P, C.POINTER(P), C.POINTER(U64)]),
    "gs_diag_filter_step2_files": (I, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(U64)]),
    "gs_sunkpos_load": (I, [C.c_char_p, C.POINTER(P)]),
    "gs_sunkpos_free": (None, [P]),
    "gs_sunkpos_n_rows": (U64, [P]),
    "gs_sunkpos_n_segs": (U32, [P]),
    "gs_sunkpos_row_seg": (P, [P]),
    "gs_sunkpos_row_pos": (P, [P]),
    "gs_sunkpos_row_contig": (P, [P]),
    "gs_sunkpos_row_start": (P, [P]),
    "gs_sunkpos_row_group": (P, [P]),
    "gs_sunkpos_seg_name": (C.c_char_p, [P, U32]),
    "gs_sunkpos_n_contigs": (U32, [P]),
    "gs_sunkpos_contig_name": (C.c_char_p, [P, U32]),
    "gs_sunkpos_row_text": (P, [P, U64, C.POINTER(U32)]),
    "gs_sunkpos_n_names": (U32, [P]),
    "gs_sunkpos_row_name": (P, [P]),
    "gs_sunkpos_name_rank": (P, [P]),
    "gs_sunkpos_name": (C.c_char_p, [P, U32]),
    "gs_sunkpos_names_long": (I, [P, C.c_char_p, C.c_int64, P]),
    "gs_sunkpos_rows_bad": (I, [P, C.c_char_p, P]),
    "gs_sunkpos_row_keys": (U32, [P, P, C.POINTER(P), C.POINTER(P)]),
    "gs_tag_hits_device": (I, [P, P, C.POINTER(P), C.POINTER(U64)]),
    "gs_locrows_create": (I, [I, U64, P, P, P, P, C.POINTER(P)]),
    "gs_locrows_from_sunkset_dev": (I, [I, P, C.POINTER(P)]),
    "gs_locrows_free": (None, [P]),
    "gs_locrows_n_rows": (U64, [P]),
    "gs_locrows_n_locs": (U32, [P]),
    "gs_locrows_key_contig": (I, [P, P]),
    "gs_rowset_create": (I, [I, U64, C.POINTER(P)]),
    "gs_rowset_free": (None, [P]),
    "gs_rowset_n": (U64, [P]),
    "gs_rowset_append_pass": (I, [P, P, P, U32, P]),
    "gs_rowset_append_host": (I, [P, U64, P, P, P, P, P, P]),
    "gs_rowset_download": (I, [P, P, P, P, P, P, P]),
    "gs_rowset_segments": (I, [P, C.POINTER(U32)]),
    "gs_rowset_diag_filter": (I, [P, U32, C.POINTER(C.c_char_p), P, P, P, P, P, C.POINTER(U64)]),
    "gs_rowset_keep_flags": (I, [P, P]),
    "gs_rowset_keep_filtered": (I, [P]),
    "gs_rowset_diag_filter_from": (I, [P, U32, C.POINTER(C.c_char_p), P, U32, P, P, P, P, C.POINTER(U64), C.POINTER(U32)]),
    "gs_rowset_bad_keys": (I, [P, P, U32, P, P, P, C.POINTER(P)]),
    "gs_rowset_bad_keys_reduced": (I, [P, P, U32, P, P, P, P, P, C.POINTER(P)]),
    "gs_device_free": (I, [I, P]),
    "gs_device_pool_trim": (I, [I]),
    "gs_selftest_inject_launch_error": (I, [I]),
    "gs_device_copy_to_host": (I, [I, P, P, U64]),
    "gs_device_copy_from_host": (I, [I, P, P, U64]),
    "gs_pairstage_from_rows": (I, [I, U64, P, P, P, U32, P, C.POINTER(P)]),
    "gs_rowset_pair_stage": (I, [P, P, P, U32, P, P, C.POINTER(P)]),
    "gs_pairstage_free": (None, [P]),
    "gs_pairstage_n_rows": (U64, [P]),
    "gs_pairstage_row_contig": (P, [P]),
    "gs_pairstage_row_name": (P, [P]),
    "gs_pairstage_row_id": (P, [P]),
    "gs_pairstage_n_regions": (U32, [P]),
    "gs_pairstage_region_contig": (P, [P]),
    "gs_pairstage_region_start": (P, [P]),
    "gs_pairstage_region_end": (P, [P]),
    "gs_pairstage_n_contigs_in": (U32, [P]),
    "gs_pairstage_contigs_in": (P, [P]),
    "gs_pairstage_counters": (P, [P]),
    "gs_diag_filter": (I, [I, P, P, P, P, P, P]),
    "gs_pair_component": (I, [I, U32, P, P, P, P, P, P]),
    "gs_key_histogram": (I, [I, U64, P, U32, P, P, P, U32]),
    "gs_union_regions": (I, [I, U64, P, P, U32, P, P, P, P]),
    "gs_sunks_define": (I, [I, I, U32, P, P, C.POINTER(P)]),
    "gs_sunks_define_device": (I, [I, I, U32, P, P, C.POINTER(P)]),
    "gs_sunkset_dev_write_files": (I, [P, C.POINTER(C.c_char_p), U32, C.c_char_p, C.c_char_p]),
    "gs_sunkset_dev_free": (None, [P]),
    "gs_sunkset_dev_n": (U64, [P]),
    "gs_sunkset_dev_n_groups": (U64, [P]),
    "gs_sunkset_dev_code": (P, [P]),
    "gs_sunkset_dev_pos": (P, [P]),
    "gs_sunkset_dev_group": (P, [P]),
    "gs_sunkset_dev_loc": (P, [P]),
    "gs_copy_to_host": (I, [I, P, P, U64]),
    "gs_sunkset_free": (None, [P]),
    "gs_sunkset_n": (U64, [P]),
    "gs_sunkset_code": (P, [P]),
    "gs_sunkset_contig": (P, [P]),
    "gs_sunkset_start": (P, [P]),
    "gs_sunkset_group": (P, [P]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m gavisunk_amd.build` "
                "(hipcc, gfx950).  gavisunk_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int) -> None:
    if status != GS_OK:
        raise GsError(status, lib().gs_last_error().decode(errors="replace"))
