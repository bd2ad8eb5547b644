"""ctypes loader + prototypes for libsnake_engine.so (include/snake_engine.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNK_LIB_PATH") or os.path.join(_HERE, "libsnake_engine.so")      # SNK_LIB_PATH: a variant build (A/B runs)

MAX_SNAKES, MAX_CELLS, MAX_NODES = 8, 361, 384
ABI_VERSION = 112        # SNK_ABI_VERSION of include/snake_engine.h these prototypes were written against


class EngineError(RuntimeError):
    pass


class SnkGameState(C.Structure):
    """mirror of snk_game_state"""
    _fields_ = [
        ("H", C.c_int32), ("W", C.c_int32), ("S", C.c_int32), ("uid", C.c_uint32),
        ("alive", C.c_uint8 * MAX_SNAKES),
        ("health", C.c_int16 * MAX_SNAKES),
        ("length", C.c_int16 * MAX_SNAKES),
        ("dir", C.c_uint8 * MAX_SNAKES),
        ("nodes", (C.c_int16 * MAX_NODES) * MAX_SNAKES),
        ("food", C.c_uint8 * MAX_CELLS),
        ("rewards", C.c_int8 * MAX_SNAKES),
        ("counters", C.c_int32 * 6),
    ]


_lib = None
vp, i32, u64, f64 = C.c_void_p, C.c_int, C.c_uint64, C.c_double

# name -> (restype, argtypes); every symbol include/snake_engine.h declares
PROTOTYPES = {
    "snk_last_error": (C.c_char_p, []),
    "snk_version": (i32, []),
    "snk_conv3x3_f16s_set_guard_word": (i32, [vp, vp, vp]),
    "snk_source_hash": (C.c_char_p, [C.c_char_p]),
    "snk_engine_create": (i32, [C.POINTER(vp), i32, i32, i32, i32, i32, f64, u64, i32]),
    "snk_engine_destroy": (i32, [vp]),
    "snk_engine_info": (i32, [vp] + [C.POINTER(i32)] * 5),
    "snk_engine_raw": (i32, [vp, C.POINTER(vp), C.POINTER(i32)]),
    "snk_engine_set_params": (i32, [vp, i32, f64]),
    "snk_engine_reset": (i32, [vp, vp, i32, vp, vp]),
    "snk_engine_clone": (i32, [vp, vp, i32, vp, vp, i32, vp]),
    "snk_engine_step": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, vp]),
    "snk_engine_step_active": (i32, [vp, vp, i32, vp, vp, vp, vp]),
    "snk_engine_alive": (i32, [vp, vp, i32, vp, vp, vp]),
    "snk_engine_ids": (i32, [vp, vp, i32, vp, vp, vp, vp, vp]),
    "snk_engine_observe": (i32, [vp, vp, i32, i32, vp, vp, vp, i32, vp]),
    "snk_engine_observe_rows": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp]),
    "snk_engine_export_sync": (i32, [vp, vp, i32, vp]),
    "snk_engine_import_sync": (i32, [vp, vp, i32, vp]),
    "snk_engine_import_at_sync": (i32, [vp, vp, i32, vp, i32]),
    "snk_engine_sum_counters_sync": (i32, [vp, vp, i32, vp]),
    "snk_compact_scratch_elems": (i32, [i32]),
    "snk_compact_flags": (i32, [vp, i32, vp, vp, vp, vp]),
    "snk_conv3x3_prepare_weights": (i32, [vp, vp, vp]),
    "snk_conv3x3_bn_f32": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "snk_conv3x3_prepare_weights_bf16": (i32, [vp, vp, vp]),
    "snk_conv3x3_prepare_weights_f16_act16": (i32, [vp, vp, vp]),
    "snk_conv3x3_bn_f16_act16_head": (i32, [vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, i32, i32, i32, vp]),
    "snk_conv3x3_bn_bf16_act16_head": (i32, [vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, i32, i32, i32, vp]),
    "snk_conv_rect_max_blocks_act16": (C.c_long, [i32, i32, i32]),
    "snk_conv_rect_plan_act16": (i32, [vp, C.c_float, C.c_float, C.c_float, i32, i32, i32, i32, C.POINTER(i32), vp, vp, vp, vp]),
    "snk_conv3x3_bn_bf16_act16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "snk_conv3x3_bn_bf16_act16_rect": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i32, i32, i32, vp]),
    "snk_stem_conv_bn_relu_bf16out": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_stem_conv_bn_relu_bf16out_rect": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "snk_conv3x3_prepare_weights_winograd": (i32, [vp, vp, vp]),
    "snk_conv3x3_bn_f32_winograd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "snk_conv3x3_prepare_weights_f16s": (i32, [vp, vp, C.c_float, vp]),
    "snk_conv3x3_bn_f16s": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "snk_conv3x3_bn_f16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "snk_conv_rect_max_blocks": (C.c_long, [i32, i32, i32]),
    "snk_conv_rect_plan": (i32, [vp, C.c_float, C.c_float, C.c_float, i32, i32, i32, i32, C.POINTER(i32), vp, vp, vp, vp]),
    "snk_stem_conv_bn_relu_f32_rect": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "snk_stem_conv_bn_relu_f16out_rect": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "snk_conv3x3_bn_f16_act16_rect": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i32, i32, i32, vp]),
    "snk_conv3x3_bn_f16s_rect": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i32, i32, i32, vp]),
    "snk_conv3x3_bn_f16s_head": (i32, [vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, i32, i32, i32, vp]),
    "snk_conv3x3_bn_f16_act16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "snk_stem_conv_bn_relu_f16out": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_head_dense_f32": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_stem_conv_bn_relu_f32": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_tt_create": (i32, [C.POINTER(vp), u64, i32]),
    "snk_tt_destroy": (i32, [vp]),
    "snk_tt_clear": (i32, [vp, vp]),
    "snk_tt_status_sync": (i32, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(i32)]),
    "snk_tt_rebuild_sync": (i32, [vp, u64, i32, i32]),
    "snk_tt_lookup_insert": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "snk_tt_find": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    "snk_tt_set_priors": (i32, [vp, vp, vp, i32, vp, vp, vp]),
    "snk_tt_read_q": (i32, [vp, vp, i32, i32, vp, vp]),
    "snk_mcts_select": (i32, [vp, vp, i32, C.c_float, vp, vp, C.c_int64, u64, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp, i32, vp, vp]),
    "snk_mcts_gather_rows": (i32, [vp, i32, vp, vp, vp, vp, vp]),
    "snk_mcts_row_active": (i32, [vp, vp, i32, i32, vp, vp]),
    "snk_mcts_retire": (i32, [vp, vp, vp, i32, i32, vp, vp, vp]),
    "snk_mcts_backup": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    "snk_mcts_terminal_backup": (i32, [vp, vp, i32, vp, vp, vp, i32, i32, vp]),
    "snk_mcts_root_moves": (i32, [vp, vp, i32, C.c_float, i32, vp, vp, C.c_int64, u64, C.c_uint32, C.c_uint32, vp, vp]),
    "snk_softermax_argmax": (i32, [vp, i32, C.c_float, vp, vp, vp]),
    "snk_engine_rewards": (i32, [vp, vp, i32, vp, vp]),
    "snk_head_f32": (i32, [vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_bn_train_partials": (i32, []),
    "snk_conv3x3_f16s_input_scale": (i32, [vp, C.c_long, vp, vp, vp]),
    "snk_conv3x3_wgrad_partials": (C.c_long, [i32, i32]),
    "snk_conv3x3_wgrad_f16s": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_bn_train_apply": (i32, [vp, vp, vp, vp, vp, C.c_long, i32, vp, vp, vp, vp]),
    "snk_bn_train_apply_head": (i32, [vp, vp, vp, vp, vp, C.c_long, vp, vp, vp, vp, vp, vp, vp]),
    "snk_bn_train_grad_apply": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_long, i32, vp, vp, vp]),
    "snk_conv3x3_prepare_weights_f16s_train": (i32, [vp, vp, vp, i32, vp, vp]),
    "snk_conv3x3_prepare_weights_f16s_train_batch": (i32, [vp, vp, vp, i32, vp]),
    "snk_stem_conv_f32": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "snk_stem_conv_f32_stats": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_conv3x3_stats_partials": (C.c_long, [i32, i32, i32]),
    "snk_conv3x3_f16s_stats": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_conv3x3_f16s_igrad_stats": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_train_deferred_bn_supported": (i32, [i32, i32]),
    "snk_stem_conv_f32_stats_deferred": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_bn_train_apply_res_deferred": (i32, [vp, vp, vp, vp, vp, vp, vp, C.c_long, vp, vp, vp, vp]),
    "snk_conv3x3_f16s_igrad_stats_masked_res_deferred": (i32, [vp] * 12 + [i32, i32, i32, vp]),
    "snk_conv3x3_f16s_stats_deferred": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_bn_train_finalize_range": (i32, [vp, f64, vp, vp, vp, vp, vp, f64, f64, vp, vp, vp, vp, vp, vp, i32, vp]),
    "snk_conv3x3_wgrad_f16s_deferred": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_conv3x3_f16s_igrad_stats_deferred": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_bn_train_grad_sums_f64_deferred": (i32, [vp, vp, vp, vp, vp, vp, C.c_long, vp, vp, vp]),
    "snk_bn_train_grad_apply_deferred": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_long, vp, vp, vp]),
    "snk_conv3x3_f16s_igrad_stats_masked_res": (i32, [vp] * 11 + [i32, i32, i32, vp]),
    "snk_stem_wgrad_partials": (C.c_long, [i32, i32, i32]),
    "snk_stem_wgrad_f32": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "snk_bn_train_sums_f64": (i32, [vp, C.c_long, vp, vp, vp, vp]),
    "snk_bn_train_finalize": (i32, [vp, f64, vp, vp, vp, vp, vp, f64, f64, vp, vp, vp, vp, i32, vp]),
    "snk_bn_train_grad_sums_f64": (i32, [vp, vp, vp, vp, vp, vp, C.c_long, i32, vp, vp, vp]),
    "snk_bn_train_grad_finalize": (i32, [vp, vp, f64, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "snk_head_conv1x1_sums": (i32, [vp, vp, C.c_long, vp, vp, vp, vp, vp]),
    "snk_head_dense_train_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f64, i32, i32, i32, vp]),
    "snk_head_dense_train_bwd_partials": (i32, [i32]),
    "snk_head_dense_train_bwd": (i32, [vp] * 10 + [f64] + [vp] * 6 + [i32, i32, i32, vp]),
    "snk_head_conv1x1_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_long, vp]),
    "snk_head_conv1x1_bwd_stats": (i32, [vp] * 15 + [C.c_long, vp]),
    "snk_adam_l2_step": (i32, [vp, vp, vp, vp, vp, C.c_long, f64, f64, f64, f64, f64, vp]),
    "snk_l2_sum": (i32, [vp, vp, C.c_long, f64, vp, vp, vp]),
    "snk_clock_probe": (i32, [vp, i32, vp]),
}


def register(protos):
    PROTOTYPES.update(protos)
    if _lib is not None:
        _bind(_lib, protos)


def _bind(L, protos):
    for name, (res, args) in protos.items():
        try:
            fn = getattr(L, name)
        except AttributeError as exc:
            raise EngineError(f"{LIB_PATH} does not export {name}; rebuild it (python -c 'import __graft_entry__ as g; g.build()')") from exc
        fn.restype = res
        fn.argtypes = args


def lib():
    """Loads the library or raises: there is deliberately no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        _bind(L, {"snk_version": PROTOTYPES["snk_version"]})
        if L.snk_version() != ABI_VERSION:      # argument lists changed between versions: a stale library would take shifted arguments
            raise EngineError(f"{LIB_PATH} reports ABI version {L.snk_version()}, these bindings are written for {ABI_VERSION}; "
                              "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
        _bind(L, PROTOTYPES)
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().snk_last_error()
        raise EngineError((msg or b"?").decode("utf-8", "replace") + f" (code {rc})")
