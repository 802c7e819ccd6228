"""ctypes binding of libilvlm_hip.so (include/ilvlm_hip.h).  The product path has no CPU or
PyTorch fallback: if the library is missing or a call fails, a RuntimeError is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ILVLM_LIB_SUFFIX: A/B builds of the library next to the shipped one (benchmarks; csrc/Makefile's stamps / ablate targets)
LIB_PATH = os.path.join(_HERE, "libilvlm_hip%s.so" % os.environ.get("ILVLM_LIB_SUFFIX", ""))

F32, BF16, FP8, FP8_BF8A = 0, 1, 2, 3
ACT_NONE, ACT_QUICKGELU, ACT_GELU_ERF, ACT_QUICKGELU_BWD, ACT_GELU_ERF_BWD = 0, 1, 2, 3, 4
POOL_MAX, POOL_MEAN, POOL_SUM = 0, 1, 2

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_long, C.c_float


class GemmEpilogue(C.Structure):
    _fields_ = [("bias", vp), ("rowbias", vp), ("residual", vp), ("aux", vp), ("alpha_ptr", vp),
                ("alpha", f32), ("act", i32), ("out_dtype", i32), ("accumulate", i32),
                ("out_group", i32), ("out_skip", i32), ("a_rowsum", vp),
                ("pool_out", vp), ("pool_seq", vp), ("pool_offs", vp), ("pool_group", i32), ("alpha_ptr2", vp),
                ("out8", vp), ("out8_scale", vp), ("out8_amax", vp), ("out8_fmt", i32),
                ("splitk_ws", vp), ("splitk_ws_bytes", i64), ("splitk_cnt", vp), ("splitk_cnt_len", i32),
                ("b_packed", vp)]


class Block(C.Structure):
    """ilvlm_block: parameters, gradient slots and geometry of one residual attention block"""
    _fields_ = ([(n, vp) for n in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "in_w", "out_w", "fc_w", "proj_w", "in_b", "out_b",
                                   "fc_b", "proj_b", "g_ln1_w", "g_ln1_b", "g_ln2_w", "g_ln2_b", "g_in_w", "g_in_b", "g_out_w",
                                   "g_out_b", "g_fc_w", "g_fc_b", "g_proj_w", "g_proj_b")] +
                [("E", i32), ("H", i32), ("causal", i32), ("dtype", i32)] +
                [(n, vp) for n in ("in_w8", "out_w8", "fc_w8", "proj_w8", "in_w8t", "out_w8t", "fc_w8t", "proj_w8t", "f8_scale",
                                   "f8_inv", "f8_amax")] + [("fp8", i32)] +
                [("splitk_ws", vp), ("splitk_ws_bytes", i64), ("splitk_cnt", vp), ("splitk_cnt_len", i32)] +
                [(n, vp) for n in ("in_wp", "out_wp", "fc_wp", "proj_wp", "in_wpt", "out_wpt", "fc_wpt", "proj_wpt")] +
                [(n, vp) for n in ("in_w8p", "out_w8p", "fc_w8p", "proj_w8p", "in_w8tp", "out_w8tp", "fc_w8tp", "proj_w8tp")])


class AugmentParams(C.Structure):
    """ilvlm_augment_params: the random draws of MOCOV2_single for one sample (made on the host)"""
    _fields_ = [("crop_top", i32), ("crop_left", i32), ("crop_h", i32), ("crop_w", i32), ("jitter", i32), ("jitter_order", i32),
                ("brightness", f32), ("contrast", f32), ("saturation", f32), ("hue", f32), ("grayscale", i32),
                ("blur_sigma", f32), ("flip", i32), ("pad_", i32)]


class TowerGrad(C.Structure):
    """ilvlm_tower_grad: where block i of a tower call puts the copies of its input gradient, and its LayerNorm workspace"""
    _fields_ = [("din_lp", vp), ("din8", vp), ("din8_scale", vp), ("din8_amax", vp), ("ln_ws", vp), ("ln_ws_blocks", i32),
                ("pad_", i32)]


BLOCK_DONE_FN = C.CFUNCTYPE(None, i32, vp)


class WgradProblem(C.Structure):
    """ilvlm_wgrad_problem: one gw[n, k] += dy^T x (gb[n] += column sums of dy) of a grouped weight-gradient launch"""
    _fields_ = [("dy", vp), ("x", vp), ("gw", vp), ("gb", vp), ("n", i32), ("k", i32), ("inv_g", vp), ("inv_x", vp)]


WGRAD_GROUP_MAX = 4


class AdamWHyper(C.Structure):
    _fields_ = [("lr", f32 * 16), ("weight_decay", f32 * 16), ("active", i32 * 16),
                ("beta1", f32), ("beta2", f32), ("eps", f32), ("step", i32)]


# name -> argtypes; every entry point declared in include/ilvlm_hip.h (tests check the .so exports them all)
SIGNATURES = {
    "ilvlm_gemm": [i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, C.POINTER(GemmEpilogue), i32, vp],
    "ilvlm_gemm_set_variant": [i32],
    "ilvlm_gemm_set_persistent": [i32, i32, i32],
    "ilvlm_gemm_set_tile_rows": [i32],
    "ilvlm_gemm_set_wgrad_tile": [i32],
    "ilvlm_gemm_set_concurrent": [i32],
    "ilvlm_gemm_get_concurrent": [],
    "ilvlm_gemm_pack_b": [i32, i32, i32, vp, i32, vp, vp],
    "ilvlm_pack_weights": [vp, vp, vp, vp, i32, vp],
    "ilvlm_wgrad_group": [i32, C.POINTER(WgradProblem), i32, i64, i32, vp],
    "ilvlm_layernorm_fwd": [vp, i32, vp, vp, vp, i32, vp, vp, i64, i32, f32, i32, i32, vp],
    "ilvlm_layernorm_bwd": [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i64, i32, i32, i32, vp, i32, vp],
    "ilvlm_layernorm_fwd_q8": [vp, i32, vp, vp, vp, i32, vp, vp, i64, i32, f32, i32, i32, vp, vp, vp, vp],
    "ilvlm_layernorm_bwd_q8": [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i64, i32, i32, i32, vp, i32, vp, vp, vp, vp],
    "ilvlm_layernorm_bwd_reduce_batched": [vp, i64, i32, i64, i32, i32, vp, vp],
    "ilvlm_attention_fwd": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "ilvlm_attention_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "ilvlm_embed_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "ilvlm_embed_bwd": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "ilvlm_patchify": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "ilvlm_cls_rows": [vp, vp, vp, i32, i32, i32, vp],
    "ilvlm_batch_sum": [vp, vp, vp, i32, i32, i32, vp],
    "ilvlm_gather_rows": [vp, vp, vp, i32, i32, i32, vp],
    "ilvlm_scatter_rows": [vp, vp, vp, i32, i32, i32, vp],
    "ilvlm_fdt_pool_fwd": [vp, vp, vp, vp, i32, i32, i32, f32, f32, i32, vp],
    "ilvlm_fdt_pool_bwd": [vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, i32, vp],
    "ilvlm_fdt_score_pool_fwd": [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, f32, vp, vp, vp],
    "ilvlm_sparsemax_fwd": [vp, vp, i32, i32, vp],
    "ilvlm_sparsemax_bwd": [vp, vp, vp, i32, i32, vp],
    "ilvlm_softmax_fwd": [vp, vp, i32, i32, vp],
    "ilvlm_softmax_bwd": [vp, vp, vp, i32, i32, vp],
    "ilvlm_sigmoid_norm_fwd": [vp, vp, vp, vp, i32, i32, vp],
    "ilvlm_sigmoid_norm_bwd": [vp, vp, vp, vp, vp, i32, i32, vp],
    "ilvlm_l2norm_fwd": [vp, vp, vp, i32, i32, f32, vp],
    "ilvlm_l2norm_bwd": [vp, vp, vp, vp, i32, i32, f32, vp],
    "ilvlm_logit_scale_fwd": [vp, vp, f32, vp],
    "ilvlm_logit_scale_bwd": [vp, vp, vp, vp, i64, vp, vp, vp, vp],
    "ilvlm_infonce_fwd": [vp, vp, i32, i32, i32, vp, vp, vp, vp],
    "ilvlm_topk_accuracy": [vp, i32, i32, i32, i32, vp, vp],
    "ilvlm_colsum": [vp, i32, vp, i64, i32, i32, vp],
    "ilvlm_fp8_quantize": [vp, i32, vp, i64, vp, vp, i32, vp],
    "ilvlm_fp8_quantize_weights": [vp, vp, vp, vp, i32, vp, vp, vp],
    "ilvlm_fp8_quantize_weights_packed": [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp],
    "ilvlm_gemm_pack_b8": [i32, i32, vp, i32, vp, vp],
    "ilvlm_fp8_scale_update": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "ilvlm_image_u8_normalize": [vp, i32, vp, vp, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp],
    "ilvlm_image_augment_scratch_floats": [i32, i32, i32],
    "ilvlm_image_augment": [vp, vp, vp, vp, vp, vp, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp],
    "ilvlm_cast_f32": [vp, vp, i32, i64, vp],
    "ilvlm_cast_to_f32": [vp, i32, vp, i64, vp],
    "ilvlm_scale": [vp, vp, f32, i64, vp],
    "ilvlm_scale_dev": [vp, vp, vp, i64, vp],
    "ilvlm_add_inplace": [vp, vp, i64, vp],
    "ilvlm_clamp": [vp, f32, f32, i64, vp],
    "ilvlm_sumsq_partials": [],
    "ilvlm_sumsq": [vp, i64, vp, vp, vp],
    "ilvlm_clip_by_norm": [vp, i64, vp, f32, vp],
    "ilvlm_adamw_step": [vp, vp, vp, vp, vp, vp, vp, vp, i32, C.POINTER(AdamWHyper), vp],
    "ilvlm_adamw_step_packed": [vp, vp, vp, vp, vp, vp, vp, vp, i32, C.POINTER(AdamWHyper), vp],
    "ilvlm_selftest_fragments": [vp, vp],
    "ilvlm_block_saved_bytes": [C.POINTER(Block), i64, i32, i32],
    "ilvlm_block_scratch_bytes": [C.POINTER(Block), i64],
    "ilvlm_block_fwd": [C.POINTER(Block), vp, vp, vp, i64, i32, i32, i32, vp, vp],
    "ilvlm_block_bwd": [C.POINTER(Block), vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp],
    "ilvlm_tower_fwd": [C.POINTER(Block), i32, vp, vp, vp, i64, i64, i32, i32, i32, vp, vp],
    "ilvlm_tower_bwd": [C.POINTER(Block), i32, C.POINTER(TowerGrad), vp, vp, vp, i64, vp, vp, vp, vp, i64, i64, i32, i32, i32, vp, i32,
                        vp, vp, BLOCK_DONE_FN, vp],
    "ilvlm_embed_packed_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "ilvlm_embed_packed_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "ilvlm_attention_packed_fwd": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp],
    "ilvlm_attention_packed_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp],
    "ilvlm_attention_fwd_q8": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "ilvlm_attention_bwd_q8": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "ilvlm_gather_packed_rows": [vp, vp, vp, vp, i32, i32, vp],
    "ilvlm_scatter_packed_rows": [vp, vp, vp, vp, i32, i32, vp],
    "ilvlm_fdt_pool_packed_fwd": [vp, vp, vp, vp, i32, i32, i32, f32, f32, i32, vp],
    "ilvlm_fdt_pool_packed_bwd": [vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, i32, vp],
    "ilvlm_tokenizer_create": [C.c_char_p, i64, C.POINTER(vp)],
    "ilvlm_tokenizer_encode": [vp, C.POINTER(C.c_char_p), i32, i32, vp, vp, vp, vp],
    "ilvlm_tokenizer_destroy": [vp],
}

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libilvlm_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C iterated-learning-for-vlm_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; load it FIRST so this library binds to the same HIP runtime instance
    # (loading /opt/rocm's copy first leaves torch and the kernels on different runtimes: "no ROCm-capable device").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.ilvlm_version.restype = i32
    lib.ilvlm_version.argtypes = []
    lib.ilvlm_last_error.restype = C.c_char_p
    lib.ilvlm_last_error.argtypes = []
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i64 if name.endswith(("_bytes", "_floats")) else i32
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("ilvlm %s failed (code %d): %s" % (what, rc, load().ilvlm_last_error().decode()))
