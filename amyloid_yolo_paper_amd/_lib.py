"""ctypes binding of libamyloid_yolo_hip.so (the C ABI declared in include/amyloid_yolo.h).

There is no CPU fallback: if the HIP library is missing or a call fails this raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libamyloid_yolo_hip.so")
ABI_VERSION = 2   # include/amyloid_yolo.h: AY_ABI_VERSION this binding's signatures (_SIGS) were written against


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "batch", "cin", "cout", "hin", "win", "hout", "wout", "ksize", "stride", "leaky", "out_f32", "cout_pad")]


class PlanOp(C.Structure):
    """ay_plan_op (include/amyloid_yolo.h)"""
    _fields_ = [("kind", C.c_int32), ("src", C.c_int32), ("src2", C.c_int32), ("res", C.c_int32), ("dst", C.c_int32),
                ("conv", ConvDesc), ("c1", C.c_int32), ("c2", C.c_int32), ("up1", C.c_int32), ("leaky2", C.c_int32),
                ("num_anchors", C.c_int32), ("num_classes", C.c_int32), ("grid", C.c_int32), ("row_offset", C.c_int32),
                ("anchors_wh", C.c_float * 12),
                ("w", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("w2", C.c_void_p), ("scale2", C.c_void_p), ("shift2", C.c_void_p)]


PLAN_INPUT, PLAN_NONE = -1, -2
OP_STEM_S2_FUSED, OP_STEM, OP_CONV, OP_RESBLOCK, OP_CONV1X1_CAT, OP_CONCAT_UPSAMPLE, OP_DECODE = 1, 2, 3, 4, 5, 6, 7


class AyError(RuntimeError):
    pass


_P, _I, _F, _SZ, _I64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64
_SIGS = {
    "ay_version": (C.c_int, []),
    "ay_last_error": (C.c_char_p, []),
    "ay_pack_conv_weights_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ay_packed_weight_bytes": (_SZ, [_I, _I, _I]),
    "ay_fold_bn": (_I, [_P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _P]),
    "ay_stem_conv_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ay_stem_s2_fused_fwd": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "ay_conv_fwd_bf16": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "ay_conv3x3_m16_fwd_bf16": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "ay_concat_upsample_bf16": (_I, [_P, _I, _I, _P, _I, _P, _I, _I, _I, _P]),
    "ay_blocked_bf16_to_nchw_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ay_blocked_f32_to_nchw_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ay_nchw_f32_to_blocked_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ay_conv_fwd_f32": (_I, [C.POINTER(ConvDesc), _P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "ay_conv_fwd_f32_valu": (_I, [C.POINTER(ConvDesc), _P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "ay_yolo_decode": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, C.POINTER(C.c_float), _I, _I, _P]),
    "ay_head_decode_fwd_bf16": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _I, _I, _I, C.POINTER(C.c_float), _P, _I, _I, _P]),
    "ay_head_decode_fwd_f16": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _I, _I, _I, C.POINTER(C.c_float), _P, _I, _I, _P]),
    "ay_xywh2xyxy": (_I, [_P, _I64, _I, _P]),
    "ay_box_iou": (_I, [_P, _I, _P, _I, _I, _I, _P, _P]),
    "ay_box_iou_pairwise": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "ay_nms_workspace_bytes": (_SZ, [_I, _I]),
    "ay_nms_filter": (_I, [_P, _I, _I, _I, _F, _P, _P, _SZ, _P]),
    "ay_nms_sort_merge": (_I, [_P, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "ay_bn_train_fwd_f32": (_I, [_P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _I, _I, _I, _P]),
    "ay_bn_train_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "ay_bias_grad_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "ay_bias_grad_f32_acc": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ay_conv_dgrad_f32": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P]),
    "ay_conv_wgrad_f32": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    "ay_add_f32": (_I, [_P, _P, _P, _SZ, _P]),
    "ay_accumulate_f32": (_I, [_P, _P, _SZ, _P]),
    "ay_copy_channels_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ay_slice_accumulate_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ay_yolo_loss_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "ay_yolo_loss_fwd_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_float), _F, _F, _P, _P, _P, _SZ, _P]),
    "ay_conv1x1_cat_fwd_bf16": (_I, [C.POINTER(ConvDesc), _P, _I, _P, _P, _P, _P, _P, _P]),
    "ay_resblock_supported": (_I, [_I]),
    "ay_plan_create": (_I, [C.POINTER(PlanOp), _I, C.POINTER(C.c_size_t), _I, _I, _I, _I, C.POINTER(C.c_void_p)]),
    "ay_plan_destroy": (None, [_P]),
    "ay_plan_workspace_bytes": (_SZ, [_P]),
    "ay_plan_value_offset": (_SZ, [_P, _I]),
    "ay_plan_forward": (_I, [_P, _P, _P, _P, _P]),
    "ay_plan_forward_timed": (_I, [_P, _P, _P, _P, C.POINTER(C.c_float), _P]),
    "ay_plan_profile_begin": (_I, [_P, C.POINTER(C.c_ubyte)]),
    "ay_plan_profile_begin_every": (_I, [_P, C.POINTER(C.c_ubyte), _I]),
    "ay_plan_profile_end": (_I, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "ay_resblock_fwd_bf16": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "ay_match_detections": (_I, [_P, _P, _I, _I, _P, _I, _F, _P, _P, _P]),
    "ay_ingest_tiles_u8": (_I, [_P, _I, _I, _I, _I, _F, _P, _P]),
    "ay_ingest_region_tiles_u8": (_I, [_P, _I, _I, _SZ, _I, _I, _I, _I, _I, _P, _P]),
    "ay_build_targets_workspace_bytes": (_SZ, [_I, _I, _I]),
    "ay_build_targets": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, C.POINTER(C.c_float), _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "ay_yolo_loss_giou_fwd_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_float), _F, _F, _P, _P, _P, _SZ, _P]),
    "ay_adam_flat": (_I, [_P, _P, _P, _P, _SZ, _F, _F, _F, _F, _I, _F, _P]),
    "ay_bn_train_fwd_bf16": (_I, [_P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ay_bn_train_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ay_bn_train_bwd_bf16_acc": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ay_bn_train_fwd_bf16_zeroed_ws": (_I, [_P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ay_bn_train_bwd_bf16_acc_zeroed_ws": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ay_accumulate_bf16": (_I, [_P, _P, _SZ, _P]),
    "ay_slice_accumulate_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ay_zero_insert_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ay_pack_dgrad_weights_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ay_pack_batch_block": (_I, []),
    "ay_pack_batch_bf16": (_I, [_P, _P, _I, _P]),
    "ay_stem_train_fwd_bf16": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ay_stem_train_stats_workspace_bytes": (C.c_size_t, []),
    "ay_stem_train_fwd_stats_bf16": (_I, [_P, _P, _P, _P, _P, C.c_size_t, _I, _I, _I, _P]),
    "ay_bn_train_apply_bf16": (_I, [_P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ay_stem_train_wgrad_workspace_bytes": (C.c_size_t, []),
    "ay_stem_train_wgrad_bf16": (_I, [_P, _P, _P, _I, _P, C.c_size_t, _I, _I, _I, _P]),
    "ay_packed_dgrad_s2_weight_bytes": (C.c_size_t, [_I, _I]),
    "ay_pack_dgrad_s2_weights_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ay_conv_dgrad_s2_bf16": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _I, _P]),
    "ay_conv_wgrad_bf16": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    "ay_conv_wgrad_bf16_acc": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P]),
    "ay_conv_wgrad_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "ay_conv_wgrad_bf16_ws": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P, C.c_size_t, _P]),
    # IEEE-half twins of the inference entry points (same signatures)
    "ay_pack_conv_weights_f16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ay_stem_conv_fwd_f16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ay_stem_s2_fused_fwd_f16": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "ay_conv_fwd_f16": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "ay_conv3x3_m16_fwd_f16": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "ay_conv1x1_cat_fwd_f16": (_I, [C.POINTER(ConvDesc), _P, _I, _P, _P, _P, _P, _P, _P]),
    "ay_resblock_fwd_f16": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "ay_blocked_f16_to_nchw_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ay_nchw_f32_to_blocked_f16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ay_stream_fence": (_I, [_P]),
    "ay_merge_detections_max_rows": (_I, []),
    "ay_merge_detections": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "ay_nms_merge": (_I, [_P, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _SZ, _P]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def lib():
    """The loaded library (loads on first use)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AyError(
                f"{LIB_PATH} is missing: build it with `python -m amyloid_yolo_paper_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.")
        l = C.CDLL(LIB_PATH)
        l.ay_version.restype = C.c_int
        have = l.ay_version()
        if have != ABI_VERSION:   # a stale .so would take e.g. ay_plan_create's out pointer for a dtype: refuse it before any call
            raise AyError(f"{LIB_PATH} exports ABI version {have}, this binding needs {ABI_VERSION}: rebuild it "
                          "(python -m amyloid_yolo_paper_amd.build --force)")
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ay_last_error()
        raise AyError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """device (or host) pointer of a torch tensor, None -> NULL"""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
