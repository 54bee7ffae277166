"""ctypes binding of libchd_spatial.so (include/chd_spatial.h).

This is the same boundary a cgo shim would use.  The library is
HIP-only: if it cannot be loaded, or no gfx950 device is usable, every entry point
of the package raises — there is no CPU path behind it.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CHD_SPATIAL_LIB: another build of the same library (A/B runs of kernel variants; channeld_amd/build.py --variant)
LIB_PATH = os.environ.get("CHD_SPATIAL_LIB") or os.path.join(HERE, "libchd_spatial.so")

OK = 0
E_CONFIG, E_INVAL, E_EXTENT, E_CENTER, E_CAPACITY, E_HANG = -1, -2, -3, -4, -5, -6
E_TOO_LARGE, E_NO_DEVICE, E_HIP, E_STATE = -8, -9, -10, -11
SHAPE_SPOTS, SHAPE_BOX, SHAPE_SPHERE, SHAPE_CONE = 1, 2, 4, 8
REC_FULL = 0x80000000
ENTITY_LOCKED = 1
WORLD_CONN_MAJOR_EMIT = 1
WORLD_CELL_MAJOR_EMIT = 2
WORLD_HANDOVER_RECIPIENTS = 4
WORLD_WIRE = 8
WORLD_OVERLAP_INTEREST = 16
WORLD_UPDATE_MASKS = 32
WORLD_ONE_WAVE_EMIT = 64
WORLD_PIPELINE_TICKS = 128
WORLD_OVERLAP_DEFERRED = 256
WORLD_GATED_OVERLAP = 512
WORLD_SEGMENTS_ONLY = 1024
WIRE_ENTITY_UPDATE, WIRE_ENTITY_FULL, WIRE_CELL_UPDATE, WIRE_CELL_FULL, WIRE_ENTITY_OBJREF = 0, 1, 2, 3, 4
HO_SRC_ONLY, HO_DST_NEW, HO_DST_KNOWN = 0, 1, 2
BROADCAST_ALL_BUT_SENDER, BROADCAST_ALL_BUT_OWNER, BROADCAST_ALL_BUT_CLIENT, BROADCAST_ALL_BUT_SERVER = 4, 8, 16, 32
BROADCAST_ADJACENT_CHANNELS = 64
MAX_DAMPING = 8
N_STAGES = 5
PROF_STAGES, PROF_RECORD_KERNEL = 0, 1
STAGE_NAMES = ("ingest", "index", "interest", "plan", "emit")

# every symbol include/chd_spatial.h declares (checked by tests/test_abi.py)
SYMBOLS = (
    "chd_create", "chd_destroy", "chd_last_error", "chd_abi_version",
    "chd_get_channel_ids", "chd_notify_decide", "chd_query_channel_ids",
    "chd_get_regions", "chd_get_adjacent_channels", "chd_server_channels",
    "chd_border_channels", "chd_world_create", "chd_world_spawn", "chd_world_despawn",
    "chd_world_set_entity_flags", "chd_subs_add", "chd_subs_remove", "chd_tick",
    "chd_tick_device", "chd_tick_fetch", "chd_sync", "chd_subs_get",
    "chd_world_get_entities", "chd_dev_alloc", "chd_dev_free", "chd_dev_upload",
    "chd_dev_download", "chd_set_profiling", "chd_set_profiling_scope", "chd_world_set_pipelining", "chd_get_tick_stats", "chd_get_tick_history",
    "chd_set_stream", "chd_world_emit_form", "chd_shard_spawn", "chd_shard_ingest", "chd_shard_import", "chd_shard_fanout",
    "chd_shard_comm_available", "chd_shard_comm_unique_id", "chd_shard_comm_init", "chd_shard_comm_destroy", "chd_shard_tick", "chd_shard_set_handover_lists", "chd_shard_set_update_senders", "chd_shard_set_update_arrivals", "chd_shard_log_spawn", "chd_shard_despawn", "chd_shard_migrate_extra_records", "chd_shard_ingest_pre", "chd_shard_ingest_post",
    "chd_shard_get_entities", "chd_shard_halo_layout", "chd_shard_interest",
    "chd_handover_recipients", "chd_adjacent_recipients", "chd_wire_set_payloads", "chd_wire_build", "chd_wire_build_info", "chd_wire_fetch",
    "chd_tick_digest", "chd_tick_fetch_segments", "chd_tick_segments", "chd_tick_segments_begin", "chd_tick_segments_end", "chd_host_alloc", "chd_host_free", "chd_subs_set_options", "chd_subs_get_options", "chd_world_set_entity_groups", "chd_world_set_handover_lists", "chd_wire_set_type_url", "chd_wire_set_merge_schema", "chd_handover_messages",
    "chd_handover_recipients_ex", "chd_handover_src_owner_unsubscribed", "chd_shard_handover_recipients", "chd_handover_variants", "chd_world_set_server_connections",
)


class ChdError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"chd error {code}: {msg}")
        self.code = code


class GridCfg(C.Structure):
    _fields_ = [
        ("grid_width", C.c_double), ("grid_height", C.c_double),
        ("world_offset_x", C.c_double), ("world_offset_z", C.c_double),
        ("grid_cols", C.c_uint32), ("grid_rows", C.c_uint32),
        ("server_cols", C.c_uint32), ("server_rows", C.c_uint32),
        ("server_interest_border_size", C.c_uint32),
        ("spatial_channel_id_start", C.c_uint32), ("entity_channel_id_start", C.c_uint32),
        ("default_fanout_interval_ms", C.c_uint32), ("default_fanout_delay_ms", C.c_int32),
        ("n_damping", C.c_uint32),
        ("damping_max_dist", C.c_uint32 * MAX_DAMPING), ("damping_interval_ms", C.c_uint32 * MAX_DAMPING),
        ("strict_load_config", C.c_uint32),
    ]


class AoiQuery(C.Structure):
    _fields_ = [
        ("shapes", C.c_uint32), ("spot_off", C.c_uint32), ("n_spots", C.c_uint32), ("n_spot_dists", C.c_uint32),
        ("box_cx", C.c_double), ("box_cz", C.c_double), ("box_ex", C.c_double), ("box_ez", C.c_double),
        ("sph_cx", C.c_double), ("sph_cz", C.c_double), ("sph_r", C.c_double),
        ("cone_cx", C.c_double), ("cone_cz", C.c_double), ("cone_dx", C.c_double), ("cone_dz", C.c_double),
        ("cone_r", C.c_double), ("cone_cos", C.c_double),
        ("_reserved", C.c_double),
    ]


assert C.sizeof(AoiQuery) == 128


class WorldCfg(C.Structure):
    _fields_ = [
        ("max_entities", C.c_uint32), ("max_subscribers", C.c_uint32), ("max_interest_cells", C.c_uint32),
        ("max_records", C.c_uint64), ("max_handovers", C.c_uint32), ("flags", C.c_uint32),
        ("wire_max_update_len", C.c_uint32), ("wire_max_full_len", C.c_uint32), ("history_depth", C.c_uint32),
        ("shard_channels", C.c_uint32),
    ]


class FanoutRec(C.Structure):
    _fields_ = [("conn", C.c_uint32), ("channel", C.c_uint32)]


class HandoverRec(C.Structure):
    _fields_ = [("entity", C.c_uint32), ("channel", C.c_uint32), ("src", C.c_uint32), ("dst", C.c_uint32),
                ("src_server", C.c_uint32), ("dst_server", C.c_uint32)]


_vp, _u32p, _f64p, _i32p, _u64p, _i64p, _u8p = (C.c_void_p,) * 7


class TickIn(C.Structure):
    _fields_ = [
        ("now_ns", C.c_int64),
        ("n_updates", C.c_uint32), ("upd_idx", _u32p), ("upd_x", _f64p), ("upd_z", _f64p), ("upd_sender", _u32p),
        ("n_cell_updates", C.c_uint32), ("cell_upd_channel", _u32p), ("cell_upd_sender", _u32p),
        ("n_queries", C.c_uint32), ("query_sub", _u32p), ("queries", _vp),
        ("spot_x", _f64p), ("spot_z", _f64p), ("spot_dist", _u32p), ("n_spots_total", C.c_uint32),
        ("upd_arrival_ns", _i64p), ("cell_upd_arrival_ns", _i64p),
        ("n_update_rounds", C.c_uint32), ("upd_round_off", _u32p),
    ]


class TickOut(C.Structure):
    _fields_ = [
        ("handovers", _vp), ("handovers_cap", C.c_uint32), ("n_handovers", C.c_uint32),
        ("n_locked_aborts", C.c_uint32),
        ("query_status", _i32p),
        ("unsub_sub", _u32p), ("unsub_channel", _u32p), ("unsub_cap", C.c_uint32), ("n_unsubs", C.c_uint32),
        ("newsub_sub", _u32p), ("newsub_channel", _u32p), ("newsub_interval_ms", _u32p),
        ("newsub_cap", C.c_uint32), ("n_newsubs", C.c_uint32),
        ("records", _vp), ("records_cap", C.c_uint64), ("n_records", C.c_uint64),
        ("conn_rec_off", _u64p), ("conn_rec_cnt", _u32p),
        ("overflow", C.c_uint32), ("history_overflow", C.c_uint32),
        ("record_masks", _u32p),
    ]


class EntityState(C.Structure):
    _fields_ = [("chan_id", C.c_uint32), ("cell", C.c_uint32), ("member", C.c_uint32), ("eflags", C.c_uint32),
                ("sender", C.c_uint32), ("hist", C.c_uint32), ("sender_prev", C.c_uint32), ("hist_prev", C.c_uint32)]


assert C.sizeof(EntityState) == 32


class TickStats(C.Structure):
    _fields_ = [
        ("stage_us", C.c_float * N_STAGES), ("total_us", C.c_float), ("emit_main_us", C.c_float),
        ("n_records", C.c_uint64), ("n_record_upper_bound", C.c_uint64),
        ("n_handovers", C.c_uint32), ("n_unsubs", C.c_uint32), ("n_pairs", C.c_uint32), ("n_deferred_records", C.c_uint32),
        ("algorithmic_bytes", C.c_uint64),
        ("n_filtered_records", C.c_uint32), ("n_deep_records", C.c_uint32),
        ("schedule", C.c_uint32), ("gate_timeouts", C.c_uint32), ("overflow", C.c_uint32), ("history_overflow", C.c_uint32),
    ]


SCHED_OVERLAP_INTEREST, SCHED_GATED, SCHED_PIPELINED, SCHED_CELL_MAJOR, SCHED_ARRIVAL_OFFSETS = 1, 2, 4, 8, 16


SUBOPT_ACCESS, SUBOPT_INTERVAL, SUBOPT_DELAY, SUBOPT_SKIP_SELF, SUBOPT_SKIP_FIRST, SUBOPT_FIELD_MASK = 1, 2, 4, 8, 16, 32
MERGE_SCHEMA_NONE, MERGE_SCHEMA_TPS_ENTITY_MOVEMENT = 0, 1
ACCESS_NONE, ACCESS_READ, ACCESS_WRITE = 0, 1, 2


class SubOptions(C.Structure):
    _fields_ = [("slot", C.c_uint32), ("channel", C.c_uint32), ("set", C.c_uint32), ("data_access", C.c_uint32),
                ("fanout_interval_ms", C.c_uint32), ("fanout_delay_ms", C.c_int32),
                ("skip_self_update_fanout", C.c_uint32), ("skip_first_fanout", C.c_uint32), ("data_field_mask", C.c_uint32)]


class HaloSeg(C.Structure):
    _fields_ = [("send_off", C.c_uint64), ("send_bytes", C.c_uint64), ("recv_off", C.c_uint64), ("recv_bytes", C.c_uint64)]


class RecordsDigest(C.Structure):
    _fields_ = [("count", C.c_uint64), ("sum", C.c_uint64), ("xor_", C.c_uint64), ("sum_masked", C.c_uint64)]


class FanoutSegment(C.Structure):
    _fields_ = [("channel", C.c_uint32), ("off", C.c_uint32), ("n_info", C.c_uint32), ("n_records", C.c_uint32)]


class SegmentsOut(C.Structure):
    _fields_ = [("segments", _vp), ("segments_cap", C.c_uint64), ("n_segments", C.c_uint64),
                ("conn_seg_off", _u32p),
                ("columns", _u32p), ("columns_cap", C.c_uint64), ("n_columns", C.c_uint64),
                ("records", _vp), ("records_cap", C.c_uint64), ("n_explicit", C.c_uint64),
                ("conn_rec_off", _u64p), ("n_records", C.c_uint64)]


class SegmentsBlock(C.Structure):  # chd_segments_block: what chd_tick_segments_end hands out (pointers into a page-locked block)
    _fields_ = [("conn_seg_off", _vp), ("conn_rec_off", _vp),
                ("segments", _vp), ("n_segments", C.c_uint64),
                ("columns", _vp), ("n_columns", C.c_uint64),
                ("records", _vp), ("n_explicit", C.c_uint64),
                ("n_records", C.c_uint64),
                ("handovers", _vp), ("n_handovers", C.c_uint32), ("n_locked_aborts", C.c_uint32),
                ("unsub_sub", _vp), ("unsub_channel", _vp), ("n_unsubs", C.c_uint32),
                ("n_newsubs", C.c_uint32),
                ("newsub_sub", _vp), ("newsub_channel", _vp), ("newsub_interval_ms", _vp),
                ("query_status", _vp), ("n_queries", C.c_uint32),
                ("overflow", C.c_uint32), ("history_overflow", C.c_uint32),
                ("reserved", C.c_uint32), ("block", _vp), ("block_bytes", C.c_uint64),
                ("wait_ms", C.c_float), ("copy_ms", C.c_float), ("device_ms", C.c_float), ("reserved2", C.c_uint32)]


SEG_FIRST, SEG_NONE, SEG_EXPLICIT = 1 << 22, 1 << 23, 1 << 24

_lib = None


def load():
    """Loads the HIP library; raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m channeld_amd.build` (hipcc, gfx950). "
            "channeld_amd has no CPU implementation."
        )
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.chd_abi_version.restype = C.c_int
    L.chd_last_error.restype = C.c_char_p
    L.chd_last_error.argtypes = [C.c_void_p]
    L.chd_create.argtypes = [P(GridCfg), C.c_int, P(C.c_void_p)]
    L.chd_destroy.argtypes = [C.c_void_p]
    L.chd_destroy.restype = None
    L.chd_sync.argtypes = [C.c_void_p]
    L.chd_get_channel_ids.argtypes = [C.c_void_p, _f64p, _f64p, C.c_uint32, _u32p]
    L.chd_notify_decide.argtypes = [C.c_void_p, _f64p, _f64p, _f64p, _f64p, C.c_uint32, _u32p, _u32p, _u8p]
    L.chd_query_channel_ids.argtypes = [C.c_void_p, _vp, C.c_uint32, _f64p, _f64p, _u32p, C.c_uint32,
                                        _u32p, _u32p, _u32p, _u32p, C.c_uint32, _i32p]
    L.chd_get_regions.argtypes = [C.c_void_p, _f64p, _f64p, _f64p, _f64p, _u32p, _u32p]
    L.chd_get_adjacent_channels.argtypes = [C.c_void_p, _u32p, C.c_uint32, _u32p, _u32p]
    L.chd_server_channels.argtypes = [C.c_void_p, C.c_uint32, _u32p, C.c_uint32, P(C.c_uint32)]
    L.chd_border_channels.argtypes = [C.c_void_p, C.c_uint32, _u32p, C.c_uint32, P(C.c_uint32)]
    L.chd_world_create.argtypes = [C.c_void_p, P(WorldCfg)]
    L.chd_world_spawn.argtypes = [C.c_void_p, C.c_uint32, _u32p, _u32p, _f64p, _f64p, _u32p, _u32p]
    L.chd_world_despawn.argtypes = [C.c_void_p, C.c_uint32, _u32p]
    L.chd_world_set_entity_flags.argtypes = [C.c_void_p, C.c_uint32, _u32p, _u32p]
    L.chd_world_set_entity_groups.argtypes = [C.c_void_p, C.c_uint32, _u32p, _u32p]
    L.chd_subs_add.argtypes = [C.c_void_p, C.c_uint32, _u32p, _u32p]
    L.chd_subs_remove.argtypes = [C.c_void_p, C.c_uint32, _u32p]
    L.chd_world_emit_form.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P(C.c_uint32)]
    L.chd_tick.argtypes = [C.c_void_p, P(TickIn), P(TickOut)]
    L.chd_tick_device.argtypes = [C.c_void_p, P(TickIn)]
    L.chd_tick_fetch.argtypes = [C.c_void_p, P(TickOut)]
    L.chd_subs_get.argtypes = [C.c_void_p, C.c_uint32, _u32p, _u32p, _i64p, _u8p, _u8p, P(C.c_uint32)]
    L.chd_world_get_entities.argtypes = [C.c_void_p, C.c_uint32, _u32p, _u32p, _u32p]
    L.chd_dev_alloc.argtypes = [C.c_void_p, C.c_uint64, P(C.c_void_p)]
    L.chd_dev_free.argtypes = [C.c_void_p, C.c_void_p]
    L.chd_dev_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.chd_dev_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.chd_set_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.chd_shard_spawn.argtypes = [C.c_void_p, C.c_uint32, _u32p, _f64p, _f64p, _u32p, _u32p]
    L.chd_shard_ingest.argtypes = [C.c_void_p, C.c_int64, _f64p, _f64p, _u8p, C.c_uint32, C.c_uint32, C.c_uint32,
                                   _vp, C.c_uint32, P(C.c_uint32)]
    L.chd_shard_halo_layout.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, P(HaloSeg), P(C.c_uint64), P(C.c_uint64)]
    L.chd_shard_import.argtypes = [C.c_void_p, _vp, C.c_uint32, C.c_uint32, _vp]
    L.chd_shard_fanout.argtypes = [C.c_void_p, _vp, C.c_uint32, P(TickIn)]
    L.chd_shard_set_handover_lists.argtypes = [C.c_void_p, C.c_uint32, _u32p, _u32p, C.c_uint32, _u32p, _u32p, C.c_uint32]
    L.chd_shard_set_update_senders.argtypes = [C.c_void_p, _vp, C.c_uint32]
    L.chd_shard_set_update_arrivals.argtypes = [C.c_void_p, _vp, C.c_uint32]
    L.chd_shard_log_spawn.argtypes = [C.c_void_p, C.c_uint32, _u32p, _f64p, _f64p]
    L.chd_shard_despawn.argtypes = [C.c_void_p, C.c_uint32, _u32p]
    L.chd_shard_migrate_extra_records.argtypes = [C.c_void_p, P(C.c_uint32)]
    L.chd_shard_ingest_pre.argtypes = [C.c_void_p, C.c_int64, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint32]
    L.chd_shard_ingest_post.argtypes = [C.c_void_p, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.POINTER(C.c_uint32)]
    L.chd_shard_comm_available.argtypes = []
    L.chd_world_set_server_connections.argtypes = [C.c_void_p, C.c_uint32, _u32p]
    L.chd_shard_comm_unique_id.argtypes = [_vp]
    L.chd_shard_comm_init.argtypes = [C.c_void_p, _vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.chd_shard_comm_destroy.argtypes = [C.c_void_p]
    L.chd_shard_tick.argtypes = [C.c_void_p, C.c_int64, _vp, _vp, _vp, C.c_uint32, P(TickIn)]
    L.chd_shard_interest.argtypes = [C.c_void_p, P(TickIn)]
    L.chd_shard_get_entities.argtypes = [C.c_void_p, _u32p, _u32p, _u32p, P(C.c_uint32)]
    L.chd_wire_set_payloads.argtypes = [C.c_void_p, C.c_int, C.c_uint32, _u32p, _u32p, _u8p]
    L.chd_wire_set_type_url.argtypes = [C.c_void_p, C.c_int, _u8p, C.c_uint32]
    L.chd_wire_set_merge_schema.argtypes = [C.c_void_p, C.c_int]
    L.chd_handover_messages.argtypes = [C.c_void_p, C.c_uint32, _u32p, _u8p, C.c_uint64, P(C.c_uint64)]
    L.chd_wire_build.argtypes = [C.c_void_p, P(C.c_uint64), P(C.c_uint64), P(C.c_uint32)]
    L.chd_wire_build_info.argtypes = [C.c_void_p, P(C.c_uint64), P(C.c_uint32)]
    L.chd_wire_fetch.argtypes = [C.c_void_p, _u64p, _u32p, _u8p, C.c_uint64]
    L.chd_handover_recipients.argtypes = [C.c_void_p, _u32p, _u32p, _u8p, C.c_uint64, P(C.c_uint64)]
    L.chd_handover_recipients_ex.argtypes = [C.c_void_p, _u32p, _u32p, _u8p, _u32p, C.c_uint64, P(C.c_uint64)]
    L.chd_handover_src_owner_unsubscribed.argtypes = [C.c_void_p, _u8p, C.c_uint32, P(C.c_uint32)]
    L.chd_shard_handover_recipients.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, _u32p, _u32p, _u8p, _u32p, _u8p, C.c_uint64, P(C.c_uint64)]
    L.chd_handover_variants.argtypes = [C.c_void_p, C.c_uint32, _u32p, _u32p, _u32p, _u8p, C.c_uint64, P(C.c_uint64)]
    L.chd_adjacent_recipients.argtypes = [C.c_void_p, C.c_uint32, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p, C.c_uint64]
    L.chd_tick_digest.argtypes = [C.c_void_p, P(RecordsDigest), _u64p]
    L.chd_tick_fetch_segments.argtypes = [C.c_void_p, P(SegmentsOut)]
    L.chd_tick_segments.argtypes = [C.c_void_p, P(TickIn), P(TickOut), P(SegmentsOut)]
    L.chd_tick_segments_begin.argtypes = [C.c_void_p, P(TickIn)]
    L.chd_tick_segments_end.argtypes = [C.c_void_p, P(SegmentsBlock)]
    L.chd_subs_set_options.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, P(SubOptions), _u8p, _i32p]
    L.chd_subs_get_options.argtypes = [C.c_void_p, C.c_uint32, _u8p, _u8p, P(C.c_uint32)]
    L.chd_host_alloc.argtypes = [C.c_void_p, C.c_uint64, P(C.c_void_p)]
    L.chd_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.chd_world_set_handover_lists.argtypes = [C.c_void_p, C.c_uint32, _u32p, _u32p, C.c_uint32, _u32p, _u32p]
    L.chd_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.chd_set_profiling_scope.argtypes = [C.c_void_p, C.c_int]
    L.chd_world_set_pipelining.argtypes = [C.c_void_p, C.c_int]
    L.chd_get_tick_stats.argtypes = [C.c_void_p, P(TickStats)]
    L.chd_get_tick_history.argtypes = [C.c_void_p, C.c_uint32, P(TickStats)]
    _lib = L
    return L


def check(ctx, rc: int):
    if rc != OK:
        msg = load().chd_last_error(ctx)
        raise ChdError(rc, msg.decode() if msg else "")
