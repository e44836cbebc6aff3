"""ctypes binding of ``libsis_hip.so`` (C ABI declared in ``include/sis_hip.h``).

This is the only place where Python touches the native library.  There is NO fallback: when the
shared object is missing, or a tensor is not on a HIP device, the wrappers raise ``RuntimeError``
(the reference's ops raise ``RuntimeError("... must be a CUDA tensor")`` in the same situations,
networks/stylegan2/op/fused_bias_act.cpp:13-14, upfirdn2d.cpp:15-16).

PyTorch is used for device memory and streams only: every wrapper takes ``torch.Tensor``s,
passes ``data_ptr()`` and the current HIP stream, and returns freshly allocated tensors, mirroring
the ownership rules of the reference extension (outputs owned by the caller, inputs borrowed and
made contiguous, kernels enqueued on the current stream, no host sync).
"""
import collections
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", os.environ.get("SIS_HIP_LIB", "libsis_hip.so"))  # (override: same-box A/B
# timing of another build of the same library, tools/build_variant.sh)

_lib = None

F32, F64, F16, BF16 = 0, 1, 2, 3
_DTYPE_CODE = {torch.float32: F32, torch.float64: F64, torch.float16: F16, torch.bfloat16: BF16}

_vp, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

_SIGNATURES = {
    "sis_version": ([], _i),
    "sis_last_error": ([], ctypes.c_char_p),
    "sis_fused_bias_act": ([_vp, _vp, _vp, _vp, _i, _i64, _i64, _i64, _i, _i, _f, _f, _vp], _i),
    "sis_upfirdn2d_out_size": ([_i] * 6, _i),
    "sis_upfirdn2d": ([_vp, _vp, _vp] + [_i] * 15 + [_vp], _i),
    "sis_pixel_norm": ([_vp, _vp, _i, _i, _vp], _i),
    "sis_equal_linear": ([_vp, _vp, _i64, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp], _i),
    "sis_truncate": ([_vp, _vp, _vp, _f, _i, _i, _vp], _i),
    "sis_modulation_batch": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp], _i),
    "sis_demod_batch": ([_vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "sis_modconv_prepack": ([_vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "sis_modconv_demod": ([_vp, _vp, _vp, _i, _i, _i, _f, _i, _vp], _i),
    "sis_modconv2d": ([_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp] + [_i] * 7 + [_vp, _vp, _i64, _vp], _i),
    "sis_head_gemm_tile": ([], _i),
    "sis_modconv_prepack_wino": ([_vp, _vp, _i, _i, _vp], _i),
    "sis_last_kernel": ([], ctypes.c_char_p),
    "sis_conv3x3_prepack": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "sis_conv3x3_prepack_both": ([_vp, _vp, _vp, _i, _i, _vp], _i),
    "sis_conv3x3_prepack_multi": ([_vp, _i, _i, _vp], _i),
    "sis_conv3x3_eligible": ([_i] * 5, _i),
    "sis_upsample_bilinear": ([_vp, _vp, _i, _i64, _i, _i, _i, _i, _i, _vp], _i),
    "sis_upsample_bilinear_strided": ([_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i64, _i, _vp], _i),
    "sis_gemm_bf16_workspace_bytes": ([_i, _i, _i], _i64),
    "sis_gemm_bf16": ([_vp] * 4 + [_i] * 8 + [_vp] * 3 + [_i] + [_vp] * 3 + [_i, _f, _i, _vp, _i64, _i, _vp], _i),
    "sis_gemm_bf16_wgrad_bias": ([_vp] * 4 + [_i] * 6 + [_vp, _i64, _i, _vp], _i),
    "sis_gemm_bf16_wgrad_multi_workspace_bytes": ([_i, _i, _i], _i64),
    "sis_gemm_bf16_wgrad_bias_multi": ([_vp] * 4 + [_i] * 6 + [_vp, _i64, _i, _vp], _i),
    "sis_gemm_bf16_batched": ([_vp] * 3 + [_i] * 9 + [_i64] * 3 + [_i, _vp, _i64, _i, _vp], _i),
    "sis_layer_norm_bwd_fused": ([_vp] * 9 + [_i, _i, _i, _i, _vp, _vp, _vp, _i, _f, _vp], _i),
    "sis_attention_fwd": ([_vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "sis_attention_bwd": ([_vp] * 6 + [_i, _i, _i, _vp], _i),
    "sis_ce_dice_workspace_floats": ([_i], _i),
    "sis_ce_dice_fwd": ([_vp] * 4 + [_i, _vp, _i, _i, _i, _vp], _i),
    "sis_ce_dice_bwd": ([_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "sis_dropout_advance": ([_vp, _vp], _i),
    "sis_dropout_bwd_cast": ([_vp, _vp, _i64, _vp, _i, _f, _vp], _i),
    "sis_layer_norm_workspace_floats": ([_i], _i),
    "sis_layer_norm_bwd_fused_partial": ([_vp] * 7 + [_i, _i, _i, _i, _vp, _vp, _vp, _i, _f, _vp], _i),
    "sis_layer_norm_bwd_parts": ([_i], _i),
    "sis_layer_norm_param_reduce_multi": ([_vp] * 5 + [_i, _vp], _i),
    "sis_column_sum_workspace_floats": ([_i], _i64),
    "sis_column_sum": ([_vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "sis_layer_norm_fwd": ([_vp] * 6 + [_i, _i, _i, _i, _f, _vp], _i),
    "sis_layer_norm_bwd": ([_vp] * 9 + [_i, _i, _i, _i, _vp], _i),
    "sis_weight_std_fwd": ([_vp, _vp, _vp, _i, _i, _i, _f, _vp], _i),
    "sis_group_norm_fwd": ([_vp] * 9 + [_i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp], _i),
    "sis_group_norm_gate_bytes": ([_i, _i, _i], _i64),
    "sis_group_norm_workspace_floats": ([_i, _i, _i], _i64),
    "sis_batch_norm_fwd": ([_vp] * 9 + [_i] * 5 + [_f, _f, _i, _vp, _vp], _i),
    "sis_batch_norm_bwd": ([_vp] * 10 + [_i] * 6 + [_vp, _vp], _i),
    "sis_group_norm_bwd": ([_vp] * 13 + [_i] * 7 + [_vp, _vp, _vp], _i),
    "sis_weight_std_bwd": ([_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp], _i),
    "sis_conv3x3_wgrad_eligible": ([_i] * 5 + [_i64], _i),
    "sis_conv3x3_wgrad": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i64, _vp], _i),
    "sis_conv3x3_wgrad_multi": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp], _i),
    "sis_conv3x3": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i64, _vp], _i),
    "sis_modconv2d_up": ([_vp] * 5 + [_i] * 6 + [_vp, _i64, _vp], _i),
    "sis_modconv_up_fir_supported": ([_i] * 6, _i),
    "sis_modconv_up_fir_prepack": ([_vp, _vp, _i, _i, _vp], _i),
    "sis_modconv2d_up_fir": ([_vp] * 5 + [_i] * 6 + [_vp], _i),
    "sis_blur_noise_act": ([_vp, _vp, _vp, _vp, _i64, _vp, _vp] + [_i] * 10 + [_vp], _i),
    "sis_to_rgb": ([_vp] * 7 + [_i] * 9 + [_f, _vp], _i),
    "sis_conv1x1_wgrad_f32_supported": ([_i] * 4, _i),
    "sis_conv1x1_wgrad_f32_workspace": ([_i] * 4, _i64),
    "sis_conv1x1_wgrad_f32": ([_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i64, _vp], _i),
    "sis_conv1x1_wgrad_f32_multi": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i64, _vp], _i),
    "sis_max_pool2d": ([_vp, _vp, _vp, _i, _i64] + [_i] * 8 + [_vp], _i),
    "sis_upsample_ce_workspace": ([_i] * 3, _i),
    "sis_upsample_ce_fwd": ([_vp] * 4 + [_i] * 6 + [_i64, _vp], _i),
    "sis_upsample_ce_bwd": ([_vp] * 4 + [_i] * 6 + [_i64, _vp], _i),
    "sis_sgd_chunk_elems": ([], _i),
    "sis_sgd_momentum": ([_vp, _i, ctypes.POINTER(_f), ctypes.POINTER(_f), _i, _f, _i, _vp], _i),
    "sis_sgd_momentum_dev": ([_vp, _i, _vp, _vp], _i),
    "sis_ema_update": ([_vp, _vp, _f, _f, _i, _i, _vp], _i),
    "sis_transpose_bf16_multi": ([_vp, _i, _i, _vp], _i),
    "sis_transpose_batched": ([_vp, _vp, _i, _i, _i, _i, _vp], _i),
    "sis_half_dilation_taps": ([_vp, _vp, _i, _i, _vp], _i),
    "sis_half_dilation_taps_bwd": ([_vp, _vp, _i, _i, _vp], _i),
    "sis_emau_supported": ([_i] * 4, _i),
    "sis_emau_workspace_floats": ([_i] * 4, _i64),
    "sis_emau_forward": ([_vp] * 5 + [_i] * 5 + [_vp], _i),
    "sis_bn_workspace_floats": ([_i, _i, _i], _i64),
    "sis_bn_stats": ([_vp] * 6 + [_i, _i, _i, _f, _f, _vp], _i),
    "sis_bn_mask_words": ([_i, _i, _i], _i64),
    "sis_bn_act_fwd": ([_vp] * 7 + [_i, _i, _i, _i, _vp, _vp], _i),
    "sis_bn_fused_supported": ([_i] * 3, _i),
    "sis_bn_fused_fwd": ([_vp] * 9 + [_i, _i, _i, _f, _f, _i, _vp, _vp], _i),
    "sis_bn_act_bwd": ([_vp] * 11 + [_i, _i, _i, _i, _vp, _vp], _i),
    "sis_kmeans_assign": ([_vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "sis_kmeans_workspace_ints": ([_i, _i], _i64),
    "sis_kmeans_assign_ws": ([_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i64, _vp], _i),
    "sis_make_image_u8": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "sis_crop_patches_u8": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "sis_assemble_max": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "sis_conv_bf16_supported": ([_i] * 6, _i),
    "sis_conv_bf16_packed_elems": ([_i] * 7, _i64),
    "sis_conv_bf16_pack": ([_vp, _vp, _i] + [_i] * 7 + [_vp], _i),
    "sis_conv_bf16": ([_vp, _vp, _vp, _vp] + [_i] * 7 + [_vp], _i),
    "sis_conv1x1_f32_supported": ([_i] * 3, _i),
    "sis_conv1x1_f32": ([_vp, _vp, _vp, _vp] + [_i] * 5 + [_vp], _i),
    "sis_conv1x1_f32_dgrad_add": ([_vp, _vp, _vp, _vp] + [_i] * 4 + [_vp], _i),
    "sis_conv_bf16_pack_both": ([_vp, _vp, _vp, _i] + [_i] * 5 + [_vp], _i),
    "sis_conv_bf16_wgrad_supported": ([_i] * 5 + [_i64], _i),
    "sis_conv_bf16_wgrad": ([_vp, _i, _vp, _vp] + [_i] * 5 + [_vp, _i64, _vp], _i),
    "sis_conv_bf16_wgrad_multi": ([_vp, _i, _vp, _vp] + [_i] * 6 + [_vp, _i64, _vp], _i),
    "sis_stem_conv_supported": ([_i] * 7, _i),
    "sis_stem_conv_packed_elems": ([], _i64),
    "sis_stem_conv_pack": ([_vp, _vp, _i, _vp], _i),
    "sis_stem_conv_fwd": ([_vp, _vp, _i, _vp, _i, _i, _i, _vp], _i),
    "sis_stem_conv_wgrad_workspace_bytes": ([_i, _i, _i], _i64),
    "sis_stem_conv_wgrad": ([_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i64, _vp], _i),
    "sis_weight_std_pack_plan": ([_i] * 4 + [_vp] * 5, _i),
    "sis_weight_std_pack_multi": ([_vp, _i, _i, _f, _vp], _i),
    "sis_weight_std_bwd_multi": ([_vp, _vp, _vp, _vp, _i, _vp], _i),
    "sis_conv1x1_bf16_wgrad_supported": ([_i] * 4 + [_i64], _i),
    "sis_conv1x1_bf16_wgrad": ([_vp, _i, _vp, _vp] + [_i] * 4 + [_vp, _i64, _vp], _i),
    "sis_conv1x1_bf16_wgrad_multi": ([_vp, _i, _vp, _vp] + [_i] * 5 + [_vp, _i64, _vp], _i),
}


def exported_symbols():
    """Every symbol include/sis_hip.h declares (checked by the CPU test-suite)."""
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libsis_hip.so is not built ({LIB_PATH}); run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C synthesis-in-style_amd/csrc`.  There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = L
    return _lib


_prof = None


def set_profiler(records):
    """bench.py hook: when ``records`` is a list, every kernel launch below is bracketed by HIP events on
    the launch stream and appended as (kernel, algorithmic_flops, algorithmic_bytes, ev0, ev1)."""
    global _prof
    _prof = records


def _launch(kernel, flops, nbytes, call):
    if _prof is None:
        return call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = call()
    e1.record()
    if kernel is None:  # convolutions: the library picks the kernel by shape / alignment and reports its name
        kernel = lib().sis_last_kernel().decode()
    _prof.append((kernel, flops, nbytes, e0, e1))
    return rc


# ---- audit of what does NOT run on this library: every place in the product that hands an operator to
# ATen / MIOpen / hipBLASLt calls ``library_call(site)``.  ``intended=True`` marks the documented exceptions (the two 3-channel
# stems); everything else is a shape the kernels declined and must stay at zero on the BASELINE configs --
# bench.py reports both counters and fails on a non-zero fallback count.
LIBRARY_CALLS = collections.Counter()
_LIBRARY_STRICT = os.environ.get("SIS_NO_LIBRARY_FALLBACK", "0") == "1"


def library_call(site, intended=False):
    LIBRARY_CALLS[("intended:" if intended else "fallback:") + site] += 1
    if _LIBRARY_STRICT and not intended:
        raise RuntimeError(f"SIS_NO_LIBRARY_FALLBACK=1: {site} fell back to the ROCm libraries")


def library_calls(reset=False):
    """{"intended": {site: n}, "fallback": {site: n}} since the last reset."""
    out = {"intended": {}, "fallback": {}}
    for key, n in LIBRARY_CALLS.items():
        kind, site = key.split(":", 1)
        out[kind][site] = n
    if reset:
        LIBRARY_CALLS.clear()
    return out


_own_kernels = None


# ---------------------------------------------------------------------------------------------------- side streams
# HIP maps streams onto a handful of hardware queues (4 on this stack), and two streams on ONE queue run their kernels one after
# the other.  Which pool stream lands beside the default stream's queue depends on what else created streams first: after an RCCL
# communicator exists, the FIRST stream torch hands out shares the default stream's queue (tools/stream_probe.py: every fourth
# one does) -- the generator's ToRGB chain then ran behind the convolutions instead of beside them (synthesis 14.29 -> 14.49 ms per
# step, tools/rccl_alive_cost.py), and a gradient exchange on such a stream would not overlap the backward at all.  So a side
# stream is CHOSEN: candidates are probed with a pair of spin kernels and the first that really runs beside its partner is kept.
_SIDE_STREAM_REJECTS = []   # candidates found on the partner's queue: kept alive so that the pool moves on
_SIDE_STREAM_LOG = []       # (device index, candidates tried, ratio of the chosen one): bench.py / tests read it


def _spin_pair_ms(main, cand, cycles):
    device = main.device
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    with torch.cuda.stream(main):
        e0.record(main)
        if cand is not None:
            cand.wait_event(e0)
        torch.cuda._sleep(cycles)
        if cand is not None:
            with torch.cuda.stream(cand):
                torch.cuda._sleep(cycles)
            main.wait_stream(cand)
        e1.record(main)
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1)


def runs_beside(main, cand, cycles=2_000_000):
    """True when kernels on ``cand`` run concurrently with kernels on ``main`` (two spin kernels take the time of one)."""
    alone = min(_spin_pair_ms(main, None, cycles) for _ in range(2))
    both = min(_spin_pair_ms(main, cand, cycles) for _ in range(2))
    return both < 1.5 * alone, both / max(alone, 1e-6)


def side_stream(device, beside=None, tries=8):
    """A stream of torch's pool whose kernels run CONCURRENTLY with those of ``beside`` (default: the current stream of
    ``device``).  SIS_STREAM_PROBE=0, or a capture in progress: the first pool stream, unprobed."""
    device = torch.device(device)
    if device.type != "cuda":
        raise ValueError("side_stream: a HIP device is required")
    if os.environ.get("SIS_STREAM_PROBE", "1") == "0" or torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device=device)
    main = beside if beside is not None else torch.cuda.current_stream(device)
    cand, ratio = None, None
    for n in range(1, tries + 1):
        cand = torch.cuda.Stream(device=device)
        ok, ratio = runs_beside(main, cand)
        if ok:
            _SIDE_STREAM_LOG.append((device.index, n, round(ratio, 2)))
            return cand
        _SIDE_STREAM_REJECTS.append(cand)
    import warnings
    warnings.warn(f"sis_hip.side_stream: none of {tries} pool streams ran beside the current stream (last ratio {ratio:.2f}); "
                  f"side-stream work will run serially")
    _SIDE_STREAM_LOG.append((device.index, tries, round(ratio, 2)))
    return cand


def own_kernel_names():
    """Names of every ``__global__`` function of csrc/*.hip (the library's own device symbols), for classifying profiler
    records into own / vendor-library kernels."""
    global _own_kernels
    if _own_kernels is None:
        import glob
        import re
        names = set()
        for path in glob.glob(os.path.join(os.path.dirname(_HERE), "csrc", "*.hip")):
            with open(path) as f:
                text = f.read()
            for m in re.finditer(r"__global__\s+(?:__launch_bounds__\s*\([^)]*\)\s*)?(?:__attribute__\s*\(\(.*?\)\)\s*)?(?:static\s+)?void\s+"
                                 r"(?:__launch_bounds__\s*\([^)]*\)\s*)?([A-Za-z_]\w*)\s*\(", text):
                names.add(m.group(1))
        _own_kernels = frozenset(names)
    return _own_kernels


def is_own_kernel(profiler_name):
    import re
    return any(tok in own_kernel_names() for tok in re.findall(r"[A-Za-z_][A-Za-z0-9_]*", profiler_name))


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed: {lib().sis_last_error().decode()}")


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_device(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")  # same text as the reference CHECK_CUDA


def dtype_code(t):
    try:
        return _DTYPE_CODE[t.dtype]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {t.dtype} (float32/float64/float16/bfloat16)") from None


# ------------------------------------------------------------------------------------------ K1


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """Same argument order as the reference pybind entry (fused_bias_act.cpp:11-21); empty tensors
    mean "absent" (fused_bias_act_kernel.cu:62-63)."""
    require_device(input, "input")
    require_device(bias, "bias")
    x = input.contiguous()
    b = bias.contiguous() if bias.numel() else None
    r = refer.contiguous() if refer.numel() else None
    if b is not None and b.dtype != x.dtype:
        b = b.to(x.dtype)
    if r is not None:
        require_device(r, "refer")
        if r.dtype != x.dtype:
            r = r.to(x.dtype)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(lib().sis_fused_bias_act(_ptr(out), _ptr(x), _ptr(b), _ptr(r), dtype_code(x), x.numel(), step_b,
                                        b.numel() if b is not None else 0, int(act), int(grad), float(alpha),
                                        float(scale), _stream()), "sis_fused_bias_act")
    return out


# ------------------------------------------------------------------------------------------ K2


def upfirdn2d_out_size(in_size, up, down, pad0, pad1, k):
    return (in_size * up + pad0 + pad1 - k + down) // down


def upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """Same signature as the reference pybind entry (upfirdn2d.cpp:12-23): input
    [major, in_h, in_w, minor], kernel [kh, kw] -> [major, out_h, out_w, minor]."""
    require_device(input, "input")
    require_device(kernel, "kernel")
    x = input.contiguous()
    k = kernel.contiguous()
    if k.dtype != x.dtype:
        k = k.to(x.dtype)
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    out_h = upfirdn2d_out_size(in_h, up_y, down_y, pad_y0, pad_y1, kh)
    out_w = upfirdn2d_out_size(in_w, up_x, down_x, pad_x0, pad_x1, kw)
    if out_h < 0 or out_w < 0:
        raise RuntimeError(f"upfirdn2d: negative output size {out_h}x{out_w}")
    out = torch.empty((major, out_h, out_w, minor), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().sis_upfirdn2d(_ptr(out), _ptr(x), _ptr(k), dtype_code(x), major, in_h, in_w, minor, kh, kw,
                                   up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, _stream()),
               "sis_upfirdn2d")
    return out


# ------------------------------------------------------------------------------ generator blocks


def _f32(t, name):
    require_device(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: the generator kernels are float32, got {t.dtype}")
    return t.contiguous()


def pixel_norm(x):
    x = _f32(x, "input")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(lib().sis_pixel_norm(_ptr(out), _ptr(x), x.shape[0], x.shape[1], _stream()), "sis_pixel_norm")
    return out


def equal_linear(x, weight, bias, scale, lr_mul, activation, row_stride=None, batch=None):
    """x is [B, in] (or any tensor whose rows are ``row_stride`` floats apart, e.g. latent[:, i])."""
    require_device(x, "input")
    w = _f32(weight, "weight")
    b = _f32(bias, "bias") if bias is not None else None
    if row_stride is None:
        x = _f32(x, "input")
        row_stride, batch = x.shape[1], x.shape[0]
    out = torch.empty((batch, w.shape[0]), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _check(lib().sis_equal_linear(_ptr(out), _ptr(x), row_stride, _ptr(w), _ptr(b), batch, w.shape[1], w.shape[0],
                                      float(scale), float(lr_mul), int(bool(activation)), _stream()),
               "sis_equal_linear")
    return out


def head_gemm_tile():
    """Output features per workgroup of the batched modulation / demodulation launches (their tables count blocks in it)."""
    return int(lib().sis_head_gemm_tile())


def modulation_batch(out_flat, latent, table, n_layers, total_blocks, scale):
    """latent [B, n_latent, dim] -> every layer's style vector, written into ``out_flat`` at the table's offsets."""
    b, n_latent, dim = latent.shape
    with torch.cuda.device(latent.device):
        _check(lib().sis_modulation_batch(_ptr(out_flat), _ptr(latent), _ptr(table), n_layers, total_blocks, b, n_latent,
                                          dim, float(scale), _stream()), "sis_modulation_batch")


def demod_batch(dscale_flat, s_flat, table, n_layers, total_blocks, batch):
    with torch.cuda.device(s_flat.device):
        _check(lib().sis_demod_batch(_ptr(dscale_flat), _ptr(s_flat), _ptr(table), n_layers, total_blocks, batch,
                                     _stream()), "sis_demod_batch")


def truncate(w, mean, psi):
    w = _f32(w, "style")
    mean = _f32(mean, "truncation_latent").reshape(-1)
    dim = w.shape[-1]
    if mean.numel() != dim:
        raise RuntimeError("truncation_latent must hold one latent vector")
    out = torch.empty_like(w)
    with torch.cuda.device(w.device):
        _check(lib().sis_truncate(_ptr(out), _ptr(w), _ptr(mean), float(psi), w.numel() // dim, dim, _stream()),
               "sis_truncate")
    return out


def modconv_prepack(weight):
    """weight: ModulatedConv2d.weight [1, Cout, Cin, k, k] -> (wpk [Cin, k*k, Cout], wsq [Cout, Cin])."""
    w = _f32(weight, "weight")
    _, cout, cin, k, _ = w.shape
    wpk = torch.empty((cin, k * k, cout), dtype=torch.float32, device=w.device)
    wsq = torch.empty((cout, cin), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _check(lib().sis_modconv_prepack(_ptr(wpk), _ptr(wsq), _ptr(w), cout, cin, k, _stream()),
               "sis_modconv_prepack")
    return wpk, wsq


def modconv_prepack_wino(weight):
    """[1, Cout, Cin, 3, 3] -> Winograd F(2x2,3x3) transformed weights [Cin, 16, Cout]."""
    w = _f32(weight, "weight")
    _, cout, cin, k, _ = w.shape
    if k != 3:
        raise RuntimeError("the Winograd transform is for 3x3 kernels")
    u = torch.empty((cin, 16, cout), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _check(lib().sis_modconv_prepack_wino(_ptr(u), _ptr(w), cout, cin, _stream()), "sis_modconv_prepack_wino")
    return u


def conv3x3_supported(x, weight, dilation=1):
    """True when the plain Winograd path takes this layer in both directions (forward and data gradient): 3x3
    kernel, float32 NCHW; a dilation d is run as d*d independent convolutions of the stride-d sub-images, so the
    constraints (W % 4 == 0, H % 2 == 0, channels % 8 == 0, tile plan fits the LDS) apply to [B*d*d, C, H/d, W/d]."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight.dim() == 4
            and tuple(weight.shape[2:]) == (3, 3) and weight.shape[1] == x.shape[1]):
        return False
    b, _, h, w = x.shape
    d = int(dilation)
    if d < 1 or h % d or w % d:
        return False
    cout, cin = weight.shape[0], weight.shape[1]
    if cout % 8 or cin % 8:
        return False
    key = (b * d * d, cin, cout, h // d, w // d)
    hit = _conv3x3_ok.get(key)
    if hit is None:
        hit = bool(lib().sis_conv3x3_eligible(*key)) and bool(lib().sis_conv3x3_eligible(key[0], cout, cin, key[3], key[4]))
        _conv3x3_ok[key] = hit
    return hit


_conv3x3_ok = {}


def conv3x3_prepack(weight, adjoint=False):
    """[Cout, Cin, 3, 3] -> Winograd-transformed weights of the forward convolution, or (adjoint) of the convolution
    that takes dL/dy to dL/dx."""
    w = _f32(weight, "weight")
    cout_w, cin_w = w.shape[0], w.shape[1]
    cout, cin = (cin_w, cout_w) if adjoint else (cout_w, cin_w)
    u = torch.empty((cin, 16, cout), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _check(lib().sis_conv3x3_prepack(_ptr(u), _ptr(w), cout, cin, int(bool(adjoint)), _stream()), "sis_conv3x3_prepack")
    return u


def conv3x3_prepack_both(weight):
    """(forward image, adjoint image) of ``conv3x3_prepack`` from one launch."""
    w = _f32(weight, "weight")
    cout, cin = w.shape[0], w.shape[1]
    u = torch.empty((cin, 16, cout), dtype=torch.float32, device=w.device)
    ua = torch.empty((cout, 16, cin), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _check(lib().sis_conv3x3_prepack_both(_ptr(u), _ptr(ua), _ptr(w), cout, cin, _stream()), "sis_conv3x3_prepack_both")
    return u, ua


class WinogradPackBank:
    """Forward and adjoint Winograd images of a list of float32 3x3 weights, refreshed by ONE launch
    (``sis_conv3x3_prepack_multi``).  ``u[i]`` [Cin,16,Cout] / ``u_adjoint[i]`` [Cout,16,Cin] keep their addresses, ``refresh()``
    rewrites their contents from the weights."""

    def __init__(self, weights):
        self.weights = [w for w in weights]
        dev = self.weights[0].device
        self.u, self.u_adjoint, rows, block = [], [], [], 0
        for w in self.weights:
            if w.dtype != torch.float32 or not w.is_contiguous() or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
                raise RuntimeError("WinogradPackBank: contiguous float32 [Cout, Cin, 3, 3] weights only")
            cout, cin = w.shape[0], w.shape[1]
            self.u.append(torch.empty((cin, 16, cout), dtype=torch.float32, device=dev))
            self.u_adjoint.append(torch.empty((cout, 16, cin), dtype=torch.float32, device=dev))
            rows.append([w.data_ptr(), self.u[-1].data_ptr(), self.u_adjoint[-1].data_ptr(), cout, cin, block])
            block += -(-cout * cin // 256)
        self.total_blocks = block
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.source_ptrs = [w.data_ptr() for w in self.weights]

    def current(self):
        return all(w.data_ptr() == ptr for w, ptr in zip(self.weights, self.source_ptrs))

    def refresh(self):
        with torch.cuda.device(self.table.device):
            _check(lib().sis_conv3x3_prepack_multi(_ptr(self.table), len(self.weights), self.total_blocks, _stream()),
                   "sis_conv3x3_prepack_multi")


def conv3x3(x, u):
    """Stride-1, padding-1 3x3 convolution of x [B,Cin,H,W] with prepacked weights u [Cin,16,Cout]."""
    x = _f32(x, "input")
    batch, cin, h, w = x.shape
    if u.shape[0] != cin:
        raise RuntimeError(f"prepacked weights are for {u.shape[0]} input channels, input has {cin}")
    cout = u.shape[2]
    out = torch.empty((batch, cout, h, w), dtype=torch.float32, device=x.device)
    ws = _workspace(x.device)
    with torch.cuda.device(x.device):
        _check(_launch(None, 2.0 * batch * cout * cin * 9 * h * w, 4.0 * (x.numel() + out.numel() + u.numel()),
                       lambda: lib().sis_conv3x3(_ptr(out), _ptr(x), _ptr(u), batch, cin, cout, h, w, _ptr(ws), ws.numel(),
                                                 _stream())), "sis_conv3x3")
    return out


def conv3x3_wgrad_supported(batch, cin, cout, h, w, min_work=0.0):
    """Capability (tile plan: channels % 64, even H / W, whole 8-tile chunks, workspace).  No size policy any more (``min_work`` =
    multiply-accumulates per tap below which the library would keep the layer): since the round-2 loop and the one-round split-K
    plan the kernel wins from 5e6 upwards (tools/bench_wgrad_small.py: 5x64->64 @16^2 0.025 vs 0.030 ms, 16x128->128 @32^2 0.056 vs
    0.079) and loses 14 us on 512-channel 4^2 / 8^2 maps (0.033 vs 0.019 ms) -- where the library's kernel adds its K slices with
    atomics and returns different bits from run to run (tools/check_conv_determinism.py); this one is deterministic."""
    if float(batch) * h * w * cin * cout < min_work:
        return False
    return bool(lib().sis_conv3x3_wgrad_eligible(batch, cin, cout, h, w, WORKSPACE_BYTES))


def conv3x3_wgrad(x, grad_output, for_param=None, defer=False):
    """dL/dw [Cout,Cin,3,3] of a stride-1, padding-1 3x3 convolution from its input x [B,Cin,H,W] and dL/dy.  ``for_param``: storage
    address of the parameter the result is the gradient of (``grad_out``).  ``defer``: as ``conv_bf16_wgrad``."""
    x = _f32(x, "input")
    gy = _f32(grad_output, "grad_output")
    batch, cin, h, w = x.shape
    cout = gy.shape[1]
    dw = grad_out(for_param, (cout, cin, 3, 3), torch.float32, x.device)
    if defer and _conv_wgrad_deferrable():
        _defer_conv_wgrad("f3", x, gy, dw, (batch, cin, cout, h, w))
        return dw
    ws = _workspace(x.device)
    with torch.cuda.device(x.device):
        _check(_launch(None, 2.0 * batch * cout * cin * 9 * h * w, 4.0 * (x.numel() + gy.numel() + dw.numel()),
                       lambda: lib().sis_conv3x3_wgrad(_ptr(dw), _ptr(x), _ptr(gy), batch, cin, cout, h, w, _ptr(ws),
                                                       ws.numel(), _stream())), "sis_conv3x3_wgrad")
    return dw


def modconv_demod(s, wsq, scale, demodulate):
    s = _f32(s, "style")
    cout, cin = wsq.shape
    out = torch.empty((s.shape[0], cout), dtype=torch.float32, device=s.device)
    with torch.cuda.device(s.device):
        _check(lib().sis_modconv_demod(_ptr(out), _ptr(s), _ptr(wsq), s.shape[0], cin, cout, float(scale),
                                       int(bool(demodulate)), _stream()), "sis_modconv_demod")
    return out


# ---- gradient arena: where a parameter's gradient is WRITTEN.  The data-parallel wrap (training/grad_exchange.py) keeps every
# gradient in flat fp32 buckets that the collectives and the fused SGD read in place.  autograd assigns a gradient that arrives
# for a parameter without one (no copy), so a weight-gradient kernel that writes its result straight into the parameter's
# bucket slice -- and returns a fresh view of it -- puts the gradient where the exchange wants it with no gather copy at all
# (the copy was 139 / 424 MB per EMANet / TransUNet iteration).  ``grad_arena_register`` maps a
# parameter's storage address to its slice; ``grad_out(key, ...)`` is what the weight-gradient bindings allocate through.
_GRAD_ARENA = {}   # parameter data_ptr -> [weakref to the parameter, flat fp32 buffer, element offset, numel, owner id, handed out]


def grad_arena_register(owner, params, flats, offsets):
    """``params[i]``'s gradient lives at ``flats[i][offsets[i] : offsets[i] + params[i].numel()]`` (fp32, contiguous)."""
    import weakref
    for prm, flat, off in zip(params, flats, offsets):
        _GRAD_ARENA[prm.data_ptr()] = [weakref.ref(prm), flat, int(off), prm.numel(), id(owner), False]


def grad_arena_reset(owner):
    """End of a backward (the owner's ``_finish``): every slice may be handed out again in the next one."""
    for ent in _GRAD_ARENA.values():
        if ent[4] == id(owner):
            ent[5] = False


def grad_arena_release(owner):
    for key in [k for k, v in _GRAD_ARENA.items() if v[4] == id(owner)]:
        del _GRAD_ARENA[key]


def grad_arena_slot(key):
    """(flat buffer, element offset, numel) registered for the parameter whose storage starts at ``key``, provided that
    parameter is alive, holds NO gradient yet (a second backward before the optimizer step must accumulate: its kernels get
    fresh tensors and autograd adds them) and its slice has not been handed out already in this backward (a parameter used by
    two nodes of the graph: the second kernel may run before autograd has accumulated the first result; it must not write
    over it); None otherwise.  A successful call marks the slice as handed out until ``grad_arena_reset``."""
    ent = _GRAD_ARENA.get(key) if key is not None else None
    if ent is None or ent[5]:
        return None
    prm = ent[0]()
    if prm is None or prm.grad is not None or prm.data_ptr() != key:
        return None
    return ent[1], ent[2], ent[3]


def _grad_arena_take(keys):
    for key in keys:
        _GRAD_ARENA[key][5] = True


def grad_out(key, shape, dtype, device):
    """Result tensor of a weight-gradient kernel for the parameter at storage address ``key`` (None / unknown: a fresh tensor):
    a NEW view of the parameter's arena slice when one is registered, so that autograd takes it as ``.grad`` as it is."""
    slot = grad_arena_slot(key)
    if slot is not None:
        flat, off, numel = slot
        n = 1
        for d in shape:
            n *= int(d)
        if n == numel and dtype == flat.dtype and device == flat.device and (flat.data_ptr() + flat.element_size() * off) % 16 == 0:
            _grad_arena_take([key])
            return flat[off:off + numel].view(shape)
    return torch.empty(shape, dtype=dtype, device=device)


def grad_out_fused(keys, rows, cols, device):
    """One [sum(rows), cols] fp32 result covering SEVERAL parameters (the ViT encoder's query | key | value product): the
    arena view when their slices lie back to back in that order, else a fresh tensor."""
    slots = [grad_arena_slot(k) for k in keys]
    if all(s is not None for s in slots):
        flat, off, _ = slots[0]
        at = off
        ok = flat.dtype == torch.float32 and flat.device == device and (flat.data_ptr() + 4 * off) % 16 == 0
        for (f, o, n), r in zip(slots, rows):
            ok = ok and f is flat and o == at and n == r * cols
            at += n
        if ok:
            _grad_arena_take(keys)
            return flat[off:at].view(sum(rows), cols)
    return torch.empty((sum(rows), cols), dtype=torch.float32, device=device)


# ---- deferred reductions.  A training step is full of tiny second-stage launches (sums of split-K slabs, of per-workgroup partial
# rows) whose results nothing reads before the optimizer or the gradient exchange: TransUNet's step had 169 of them, 1.6 ms of
# 23.7 -- mostly kernel boundaries.  Inside an autograd backward a binding may therefore DEFER such a reduction: the
# first-stage kernel runs, its partial results stay in their buffer, and the job joins a queue that ONE launch per kind drains
# (``flush_deferred``) at the end of the backward (an engine callback), before a data-parallel bucket leaves, before the fused
# SGD step, or on request.  Until the flush the result tensors (``.grad`` of the parameters) hold no valid data, so a call site
# may only defer a gradient whose parameter has NO gradient yet (``defer=`` of the bindings: with one in place autograd adds the
# new tensor to it immediately -- gradient accumulation over two backwards keeps the undeferred path).  Outside a backward (tests
# calling a binding directly) nothing is deferred.  SIS_DEFER_REDUCES=0 switches it off (A/B runs).
_DEFER = os.environ.get("SIS_DEFER_REDUCES", "1") != "0"
_deferred = {"ln": [], "wgrad": {}, "conv": {}}   # wgrad: (tokens, out, in, lda, ldb, device) -> [(grad, x, dw, db addresses, storages)]
_deferred_task = [None]         # the running backward has its end-of-backward callback queued (reset by the callback itself:
                                # graph task ids are not unique across backwards)

_defer_blockers = set()   # ids of wrappers that read a gradient the moment autograd hands it over (torch's DistributedDataParallel)


def block_deferral(owner, blocked=True):
    """No deferred completion of gradients while ``owner`` lives: for consumers that read ``.grad`` inside the backward."""
    (_defer_blockers.add if blocked else _defer_blockers.discard)(id(owner))


def deferring():
    """True inside an autograd backward with deferral switched on; queues the end-of-backward flush on first use per backward."""
    if not _DEFER or _defer_blockers:
        return False
    task = torch._C._current_graph_task_id()
    if task < 0:
        return False
    if _deferred_task[0] != task:
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
        _deferred_task[0] = task
    return True


def _end_of_backward():
    _deferred_task[0] = None
    flush_deferred()


def _hold(*tensors):
    """What a queued job keeps of its tensors: their storages, not the tensors.  A result tensor that a job referenced would
    reach autograd's AccumulateGrad with a use count of two, and a gradient that is not uniquely referenced is CLONED there --
    before the flush has written it.  The storage keeps the memory alive without counting as a reference to the tensor."""
    return tuple(t.untyped_storage() for t in tensors if t is not None)


def deferred_pending():
    return len(_deferred["ln"]) + sum(len(v) for v in _deferred["wgrad"].values()) + sum(len(v) for v in _deferred["conv"].values())


_DEFER_WGRAD = os.environ.get("SIS_DEFER_WGRAD", "1") != "0"   # 0: the encoder's weight-gradient GEMMs where their operands appear
_defer_wgrad_blockers = set()   # ids of data-parallel wrappers with more than one rank: their buckets leave DURING the backward,
                                # overlapped with it -- a gradient held back to its end would put the exchange behind it


def block_wgrad_deferral(owner, blocked=True):
    (_defer_wgrad_blockers.add if blocked else _defer_wgrad_blockers.discard)(id(owner))


def deferred_wgrad_pending():
    return any(_deferred["wgrad"].values()) or any(_deferred["conv"].values())


def _defer_conv_wgrad(kind, x, grad_output, dw, dims):
    """Queues a bf16 convolution's weight gradient (kind 3: 3x3, 1: 1x1) for the batched launch of ``flush_deferred``: layers of
    one shape (the trunk's repeated bottleneck units) share one tile launch and one reduction launch, planned for their joint
    workgroup count.  ``dw`` is valid after the flush only; the caller guarantees that nothing reads it before."""
    key = (kind,) + tuple(dims) + (dw.dtype, x.device)
    _deferred["conv"].setdefault(key, []).append((x.data_ptr(), grad_output.data_ptr(), dw.data_ptr(), _hold(x, grad_output, dw)))


def _conv_wgrad_deferrable():
    return _DEFER_WGRAD and not _defer_wgrad_blockers and not torch.is_grad_enabled() and deferring()


def _flush_conv_wgrads():
    queued, _deferred["conv"] = _deferred["conv"], {}
    for key, jobs in queued.items():
        kind, dims, dtype, device = key[0], key[1:-2], key[-2], key[-1]
        n = len(jobs)
        arr = ctypes.c_void_p * n
        xs, gys, dws = arr(*[j[0] for j in jobs]), arr(*[j[1] for j in jobs]), arr(*[j[2] for j in jobs])
        ws = _workspace(device)
        with torch.cuda.device(device):
            if kind == "f3":   # fp32 Winograd weight gradient (EMANet)
                b, cin, cout, h, w = dims
                _check(_launch(None, 2.0 * b * cout * cin * 9 * h * w * n, 0.0,
                               lambda: lib().sis_conv3x3_wgrad_multi(dws, xs, gys, n, b, cin, cout, h, w, _ptr(ws), ws.numel(), _stream())),
                       "sis_conv3x3_wgrad_multi")
            elif kind == "f1":   # fp32 1x1 weight gradient (EMANet)
                b, cin, cout, hw = dims
                _check(_launch("conv1x1_wgrad_f32_kernel", 2.0 * b * cout * cin * hw * n, 0.0,
                               lambda: lib().sis_conv1x1_wgrad_f32_multi(dws, gys, xs, n, b, cin, cout, hw, _ptr(ws), ws.numel(), _stream())),
                       "sis_conv1x1_wgrad_f32_multi")
            elif kind == 3:
                b, cin, cout, h, w = dims
                _check(_launch(None, 2.0 * b * cout * cin * 9 * h * w * n, 0.0,
                               lambda: lib().sis_conv_bf16_wgrad_multi(dws, _DTYPE_CODE[dtype], xs, gys, n, b, cin, cout, h, w, _ptr(ws),
                                                                       ws.numel(), _stream())), "sis_conv_bf16_wgrad_multi")
            else:
                b, cin, cout, pixels = dims
                _check(_launch(None, 2.0 * b * cout * cin * pixels * n, 0.0,
                               lambda: lib().sis_conv1x1_bf16_wgrad_multi(dws, _DTYPE_CODE[dtype], xs, gys, n, b, cin, cout, pixels, _ptr(ws),
                                                                          ws.numel(), _stream())), "sis_conv1x1_bf16_wgrad_multi")


def defer_wgrad_bias(grad, x, dw=None):
    """Queues the weight / bias gradient of a Linear layer (``gemm_bf16_wgrad_bias``'s operands) for the batched launch of
    ``flush_deferred`` -- one pointer-table GEMM per Linear shape for all queued layers, each contracting its whole token axis per
    tile (no split-K slabs, no reduction launches: 2.61 -> 2.02 ms for the twelve blocks of ViT-B at 8 192 tokens,
    tools/bench_wgrad_multi.py).  Returns (dw, db): allocated now, VALID ONLY AFTER THE FLUSH (the caller checked that the
    parameters hold no gradient yet, see the comment above).  None when the operands do not qualify (the caller then multiplies
    on the spot)."""
    if not (_DEFER_WGRAD and grad.dim() == 2 and x.dim() == 2 and grad.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
            and grad.stride(1) == 1 and x.stride(1) == 1 and grad.shape[0] == x.shape[0] and grad.stride(0) % 8 == 0
            and x.stride(0) % 8 == 0 and grad.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and grad.shape[1] % 4 == 0
            and x.shape[1] % 4 == 0 and not _defer_wgrad_blockers and deferring()):
        return None
    k, m = grad.shape
    n = x.shape[1]
    if dw is None:
        dw = torch.empty((m, n), dtype=torch.float32, device=grad.device)
    elif tuple(dw.shape) != (m, n) or dw.dtype != torch.float32 or not dw.is_contiguous():
        raise RuntimeError("defer_wgrad_bias: dw must be a contiguous float32 [out, in] tensor")
    db = torch.empty(m, dtype=torch.float32, device=grad.device)
    key = (k, m, n, grad.stride(0), x.stride(0), grad.device)
    _deferred["wgrad"].setdefault(key, []).append((grad.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), _hold(grad, x, dw, db)))
    return dw, db


def flush_deferred():
    """Runs every queued reduction (one launch per 32 jobs) and weight gradient (one launch per Linear shape and 16 jobs).
    Cheap when nothing is queued."""
    if _deferred["conv"]:
        _flush_conv_wgrads()
    if _deferred["wgrad"]:
        queued, _deferred["wgrad"] = _deferred["wgrad"], {}
        for key, jobs in queued.items():
            _wgrad_multi_raw([j[:4] for j in jobs], key)
    jobs = _deferred["ln"]
    if jobs:
        _deferred["ln"] = []
        n = len(jobs)
        vp, ci = ctypes.c_void_p * n, ctypes.c_int * n
        with torch.cuda.device(jobs[0][5]):
            _check(lib().sis_layer_norm_param_reduce_multi(vp(*[j[0] for j in jobs]), vp(*[j[1] for j in jobs]), vp(*[j[2] for j in jobs]),
                                                           ci(*[j[3] for j in jobs]), ci(*[j[4] for j in jobs]), n, _stream()),
                   "sis_layer_norm_param_reduce_multi")


WORKSPACE_BYTES = int(os.environ.get("SIS_WORKSPACE_MB", "128")) << 20  # split-K slabs (per device)
_workspaces = {}


def _workspace(device):
    """Per-device scratch for split-K partial sums (stream-ordered reuse: one stream per device at a time)."""
    ws = _workspaces.get(device)
    if ws is None:
        ws = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=device)
        _workspaces[device] = ws
    return ws


def _noise_args(noise, batch, h, w):
    if noise is None:
        return None, 0
    noise = _f32(noise, "noise")
    if noise.numel() == h * w:
        return noise, 0
    if noise.numel() == batch * h * w:
        return noise, h * w
    raise RuntimeError(f"noise shape {tuple(noise.shape)} does not broadcast over [{batch}, C, {h}, {w}]")


def _out_or_new(out, shape, device, what):
    """``out``: a caller-provided contiguous float32 result tensor (e.g. a batch slice of a larger one) or None."""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != device:
        raise RuntimeError(f"{what}: out must be a contiguous float32 tensor of shape {tuple(shape)} on {device}")
    return out


def modconv2d(x, wpk, s, dscale, ksize, noise=None, noise_weight=None, bias=None, fuse_act=False, wino_u=None, out=None):
    x = _f32(x, "input")
    batch, cin, h, w = x.shape
    cout = wpk.shape[2]
    noise, nbs = _noise_args(noise, batch, h, w)
    out = _out_or_new(out, (batch, cout, h, w), x.device, "modconv2d")
    ws = _workspace(x.device)
    wino = wino_u is not None and ksize == 3 and h % 2 == 0 and w % 2 == 0 and cin % 8 == 0 and cout % 4 == 0
    with torch.cuda.device(x.device):
        _check(_launch(None,
                       2.0 * batch * cout * cin * ksize * ksize * h * w,
                       4.0 * (x.numel() + out.numel() + wpk.numel()),
                       lambda: lib().sis_modconv2d(_ptr(out), _ptr(x), _ptr(wpk), _ptr(s), _ptr(dscale), _ptr(noise),
                                                   nbs, _ptr(noise_weight), _ptr(bias), batch, cin, cout, h, w, ksize,
                                                   int(bool(fuse_act)), _ptr(wino_u), _ptr(ws), ws.numel(), _stream())),
               "sis_modconv2d")
    return out


_UP_FIR = os.environ.get("SIS_UP_FIR", "1") != "0"   # 0: up-convolutions on the 4-phase gather kernel only (A/B runs)


def modconv_prepack_up_fir(weight):
    """[1, Cout, Cin, 3, 3] parameter -> the 16 transformed planes of the fast-FIR up-convolution, [Cin, 8 plane pairs, Cout, 2]."""
    w = _f32(weight, "weight")
    if w.dim() == 5:
        w = w[0]
    cout, cin = w.shape[0], w.shape[1]
    u = torch.empty((cin, 8, cout, 2), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _check(lib().sis_modconv_up_fir_prepack(_ptr(u), _ptr(w), cout, cin, _stream()), "sis_modconv_up_fir_prepack")
    return u


def modconv2d_up(x, wpk, s, dscale, padded_rows=False, fir_u=None):
    """Transposed stride-2 modulated conv -> [B, Cout, 2H+1, 2W+1].  ``padded_rows=True`` returns the buffer with
    rows padded to 2W+4 floats ([..., 2H+1, 2W+4], first 2W+1 columns meaningful) for sis_blur_noise_act.  ``fir_u``
    (modconv_prepack_up_fir) selects the fast-FIR kernel where it applies (padded rows, maps of 32 x 32 and larger)."""
    x = _f32(x, "input")
    batch, cin, h, w = x.shape
    cout = wpk.shape[2]
    row = 2 * w + 4 if padded_rows else 2 * w + 1
    out = torch.empty((batch, cout, 2 * h + 1, row), dtype=torch.float32, device=x.device)
    if fir_u is not None and _UP_FIR and padded_rows and lib().sis_modconv_up_fir_supported(batch, cin, cout, h, w, row):
        with torch.cuda.device(x.device):
            # (FLOPs recorded in the direct 9-multiplies-per-position count of SURVEY.md 8(d); the kernel executes 25 / 36 of it)
            _check(_launch(None, 2.0 * batch * cout * cin * 9 * h * w, 4.0 * (x.numel() + out.numel() + fir_u.numel()),
                           lambda: lib().sis_modconv2d_up_fir(_ptr(out), _ptr(x), _ptr(fir_u), _ptr(s), _ptr(dscale), batch, cin, cout,
                                                              h, w, row, _stream())), "sis_modconv2d_up_fir")
        return out
    ws = _workspace(x.device)
    with torch.cuda.device(x.device):
        _check(_launch(None, 2.0 * batch * cout * cin * 9 * h * w,
                       4.0 * (x.numel() + out.numel() + wpk.numel()),
                       lambda: lib().sis_modconv2d_up(_ptr(out), _ptr(x), _ptr(wpk), _ptr(s), _ptr(dscale), batch, cin,
                                                      cout, h, w, row, _ptr(ws), ws.numel(), _stream())),
               "sis_modconv2d_up")
    return out


def blur_noise_act(x, taps, pad, noise=None, noise_weight=None, bias=None, fuse_act=False, in_w=None):
    """``in_w``: meaningful width when the rows of ``x`` are padded (x.shape[3] is then the row stride)."""
    x = _f32(x, "input")
    taps = _f32(taps, "kernel")
    batch, ch, ih, row_stride = x.shape
    iw = row_stride if in_w is None else in_w
    kh, kw = taps.shape
    oh, ow = ih + pad[0] + pad[1] - kh + 1, iw + pad[0] + pad[1] - kw + 1
    noise, nbs = _noise_args(noise, batch, oh, ow)
    out = torch.empty((batch, ch, oh, ow), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(_launch("blur_rows_kernel", 0.0, 4.0 * (batch * ch * ih * iw + out.numel()),
                       lambda: lib().sis_blur_noise_act(_ptr(out), _ptr(x), _ptr(taps), _ptr(noise), nbs,
                                                        _ptr(noise_weight), _ptr(bias), batch, ch, ih, iw, row_stride,
                                                        kh, kw, pad[0], pad[1], int(bool(fuse_act)), _stream())),
               "sis_blur_noise_act")
    return out


def to_rgb(x, weight, s, bias, scale, skip=None, taps=None, pad=(0, 0), out=None):
    x = _f32(x, "input")
    w = _f32(weight, "weight")
    batch, cin, h, wd = x.shape
    cout = w.numel() // cin
    kh = kw = 0
    if skip is not None:
        skip = _f32(skip, "skip")
        taps = _f32(taps, "kernel")
        kh, kw = taps.shape
    out = _out_or_new(out, (batch, cout, h, wd), x.device, "to_rgb")
    with torch.cuda.device(x.device):
        _check(_launch("to_rgb_kernel", 0.0, 4.0 * (x.numel() + out.numel() + (skip.numel() if skip is not None else 0)),
                       lambda: lib().sis_to_rgb(_ptr(out), _ptr(x), _ptr(w), _ptr(s), _ptr(bias), _ptr(skip),
                                                _ptr(taps), batch, cin, cout, h, wd, kh, kw, pad[0], pad[1],
                                                float(scale), _stream())), "sis_to_rgb")
    return out


# ------------------------------------------------------------------------------ segmentation training


def upsample_ce_fwd(logits, labels, out_size, ignore_index):
    """loss[B] of bilinear(align_corners) + log-softmax + NLL + spatial mean; labels int64 [B, H, W]."""
    x = _f32(logits, "logits")
    require_device(labels, "labels")
    if labels.dtype != torch.int64:
        raise RuntimeError(f"labels must be int64, got {labels.dtype}")
    lab = labels.contiguous()
    b, c, h, w = x.shape
    oh, ow = out_size
    if tuple(lab.shape) != (b, oh, ow):
        raise RuntimeError(f"labels shape {tuple(lab.shape)} != {(b, oh, ow)}")
    L = lib()
    ws = torch.empty(L.sis_upsample_ce_workspace(b, oh, ow), dtype=torch.float32, device=x.device)
    loss = torch.empty(b, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(L.sis_upsample_ce_fwd(_ptr(loss), _ptr(ws), _ptr(x), _ptr(lab), b, c, h, w, oh, ow, int(ignore_index),
                                     _stream()), "sis_upsample_ce_fwd")
    return loss


def upsample_ce_bwd(grad_loss, logits, labels, out_size, ignore_index):
    x = _f32(logits, "logits")
    g = _f32(grad_loss, "grad_loss")
    lab = labels.contiguous()
    b, c, h, w = x.shape
    gx = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(lib().sis_upsample_ce_bwd(_ptr(gx), _ptr(g), _ptr(x), _ptr(lab), b, c, h, w, out_size[0], out_size[1],
                                         int(ignore_index), _stream()), "sis_upsample_ce_bwd")
    return gx


def sgd_chunk_elems():
    return lib().sis_sgd_chunk_elems()


def sgd_momentum(table, n_chunks, lrs, wds, momentum, first_step):
    n = len(lrs)
    arr = _f * n
    with torch.cuda.device(table.device):
        _check(lib().sis_sgd_momentum(_ptr(table), n_chunks, arr(*lrs), arr(*wds), n, float(momentum),
                                      int(bool(first_step)), _stream()), "sis_sgd_momentum")


def sgd_momentum_dev(table, n_chunks, hyper):
    """Capturable variant: lr[4], wd[4], momentum are read from the device tensor ``hyper`` (float32[9])."""
    with torch.cuda.device(table.device):
        _check(lib().sis_sgd_momentum_dev(_ptr(table), n_chunks, _ptr(hyper), _stream()), "sis_sgd_momentum_dev")


def ema_update(mu, mu_batch, momentum):
    """In place: mu <- mu*m + mean_b(mu_batch)*(1-m); (1-m) is formed in double like the reference's Python."""
    require_device(mu, "mu")
    mb = _f32(mu_batch, "mu_batch")
    if not mu.is_contiguous() or mu.dtype != torch.float32:
        raise RuntimeError("mu must be a contiguous float32 buffer")
    n = mu.numel()
    with torch.cuda.device(mu.device):
        _check(lib().sis_ema_update(_ptr(mu), _ptr(mb), float(momentum), float(1 - momentum), mb.numel() // n, n,
                                    _stream()), "sis_ema_update")
    return mu


def half_dilation_taps(weight):
    """[4 cout, 4 cin] matrix of a 3x3 convolution whose dilation is half the image side (see include/sis_hip.h)."""
    require_device(weight, "weight")
    w = _f32(weight, "weight")
    cout, cin = w.shape[0], w.shape[1]
    taps = torch.empty((4 * cout, 4 * cin), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _check(lib().sis_half_dilation_taps(_ptr(taps), _ptr(w), cout, cin, _stream()), "sis_half_dilation_taps")
    return taps


def half_dilation_taps_bwd(grad_taps, cout, cin, for_param=None):
    g = _f32(grad_taps, "grad_taps")
    dw = grad_out(for_param, (cout, cin, 3, 3), torch.float32, g.device)
    with torch.cuda.device(g.device):
        _check(lib().sis_half_dilation_taps_bwd(_ptr(dw), _ptr(g), cout, cin, _stream()), "sis_half_dilation_taps_bwd")
    return dw


def emau_supported(x, mu):
    """x [b, c, h, w] (or [b, c, n]) float32 on a HIP device, mu [1, c, 64]."""
    if not (x.is_cuda and x.dtype == torch.float32 and mu.dtype == torch.float32 and x.dim() in (3, 4) and mu.dim() == 3):
        return False
    b, c = x.shape[0], x.shape[1]
    n = x.numel() // max(b * c, 1)
    return mu.shape[0] == 1 and mu.shape[1] == c and bool(lib().sis_emau_supported(b, c, n, mu.shape[2]))


def emau_forward(x, mu, stages):
    """The EM rounds + reconstruction of EMAU.forward (networks/ema_net/network.py:229-247) -> (relu(mu z^T) shaped like x,
    mu [b, c, k]); no autograd graph (the reference runs the rounds under no_grad and the reconstruction only sees their
    results)."""
    require_device(x, "x")
    x, m0 = _f32(x, "x"), _f32(mu, "mu")
    b, c = x.shape[0], x.shape[1]
    n, k = x.numel() // (b * c), m0.shape[2]
    out = torch.empty_like(x)
    mu_out = torch.empty((b, c, k), dtype=torch.float32, device=x.device)
    ws = torch.empty(lib().sis_emau_workspace_floats(b, c, n, k), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(_launch("emau_kernels", 2.0 * b * c * n * k * (2 * stages + 1), 4.0 * (2 * x.numel() + (2 * stages + 1) * (b * n * k + b * c * k)),
                       lambda: lib().sis_emau_forward(_ptr(out), _ptr(mu_out), _ptr(x), _ptr(m0), _ptr(ws), b, c, n, k, int(stages),
                                                      _stream())), "sis_emau_forward")
    return out, mu_out


# ------------------------------------------------------------------------------ dataset-loop neighbours


def kmeans_assign(x, centres):
    """[B,C,H,W] activations, [K,C] centres -> int64 [B,H,W] nearest-centre ids (FactorCatalog.predict)."""
    x = _f32(x, "input")
    c = _f32(centres, "cluster_centers")
    b, ch, h, w = x.shape
    if c.shape[1] != ch:
        raise RuntimeError(f"centres have {c.shape[1]} channels, activations {ch}")
    labels = torch.empty((b, h, w), dtype=torch.int64, device=x.device)
    ws = torch.empty(lib().sis_kmeans_workspace_ints(b, h * w), dtype=torch.int32, device=x.device)   # list of the pixels the fast pass leaves open
    with torch.cuda.device(x.device):
        _check(_launch("kmeans_assign_kernel", 3.0 * b * ch * h * w * c.shape[0], 4.0 * x.numel() + 8.0 * labels.numel(),
                       lambda: lib().sis_kmeans_assign_ws(_ptr(labels), _ptr(x), _ptr(c), b, ch, h * w, c.shape[0], _ptr(ws), ws.numel(),
                                                          _stream())), "sis_kmeans_assign")
    return labels


def make_image_u8(x):
    """[B,C,H,W] float32 in [-1,1] -> uint8 [B,H,W,C] on the device."""
    x = _f32(x, "image")
    b, ch, h, w = x.shape
    out = torch.empty((b, h, w, ch), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().sis_make_image_u8(_ptr(out), _ptr(x), b, ch, h * w, _stream()), "sis_make_image_u8")
    return out


# ------------------------------------------------------------------------------ fp32 pointwise convolution (matrix cores, NCHW)


def conv1x1_f32_supported(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and x.data_ptr() % 16 == 0
            and bool(lib().sis_conv1x1_f32_supported(weight.shape[1], weight.shape[0], x.shape[2] * x.shape[3])))


def conv1x1_f32(x, weight, bias=None, data_gradient=False):
    """1x1 stride-1 fp32 convolution of x [B,Cin,H,W] with weight [Cout,Cin,1,1] (+ bias), or with ``data_gradient`` the
    gradient w.r.t. the input from x = dL/dy [B,Cout,H,W]."""
    x = _f32(x, "input")
    w = _f32(weight, "weight")
    b, cx, h, wd = x.shape
    cout, cin = w.shape[0], w.shape[1]
    if cx != (cout if data_gradient else cin):
        raise RuntimeError(f"conv1x1_f32: input has {cx} channels, weight is {cout}x{cin}")
    out = torch.empty((b, cin if data_gradient else cout, h, wd), dtype=torch.float32, device=x.device)
    if bias is not None:
        bias = _f32(bias, "bias")
    with torch.cuda.device(x.device):
        _check(_launch(None, 2.0 * b * cout * cin * h * wd, 4.0 * (x.numel() + out.numel() + w.numel()),
                       lambda: lib().sis_conv1x1_f32(_ptr(out), _ptr(x), _ptr(w), _ptr(bias), b, cin, cout, h * wd,
                                                     int(bool(data_gradient)), _stream())), "sis_conv1x1_f32")
    return out


def conv1x1_f32_dgrad_add(grad_output, weight, skip_grad):
    """dL/dx of a 1x1 stride-1 fp32 convolution plus ``skip_grad`` (same shape as the result), one launch."""
    g = _f32(grad_output, "grad_output")
    w = _f32(weight, "weight")
    skip = _f32(skip_grad, "skip_grad")
    b, cout, h, wd = g.shape
    cin = w.shape[1]
    if tuple(skip.shape) != (b, cin, h, wd):
        raise RuntimeError("conv1x1_f32_dgrad_add: skip_grad must have the input's shape")
    out = torch.empty((b, cin, h, wd), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _check(_launch(None, 2.0 * b * cout * cin * h * wd, 4.0 * (g.numel() + 2 * out.numel() + w.numel()),
                       lambda: lib().sis_conv1x1_f32_dgrad_add(_ptr(out), _ptr(g), _ptr(w), _ptr(skip), b, cin, cout, h * wd, _stream())),
               "sis_conv1x1_f32_dgrad_add")
    return out


# ------------------------------------------------------------------------------ bf16 convolutions (matrix cores, NCHW)


_conv_bf16_ok = {}


def conv_bf16_supported(cin, cout, h, w, ksize, stride):
    key = (cin, cout, h, w, ksize, stride)
    hit = _conv_bf16_ok.get(key)
    if hit is None:
        hit = _conv_bf16_ok[key] = bool(lib().sis_conv_bf16_supported(*key))
    return hit


def conv_bf16_pack(weight, h, w, stride=1, adjoint=False):
    """weight [Cout, Cin, k, k] (float32 or bfloat16) -> the kernel's packed LDS image (bf16) for inputs of h x w;
    ``adjoint``: the packing of the data-gradient convolution of a stride-1 layer."""
    require_device(weight, "weight")
    wt = weight.contiguous()
    if wt.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"conv_bf16_pack: weights must be float32 or bfloat16, got {wt.dtype}")
    cout, cin, k, k2 = wt.shape
    n = lib().sis_conv_bf16_packed_elems(cin, cout, h, w, k, stride, int(bool(adjoint)))
    if k != k2 or n < 0:
        raise RuntimeError(f"conv_bf16_pack: unsupported layer {cin}->{cout} k{k}x{k2} s{stride}")
    packed = torch.empty(n, dtype=torch.bfloat16, device=wt.device)
    with torch.cuda.device(wt.device):
        _check(lib().sis_conv_bf16_pack(_ptr(packed), _ptr(wt), _DTYPE_CODE[wt.dtype], cin, cout, h, w, k, stride,
                                        int(bool(adjoint)), _stream()), "sis_conv_bf16_pack")
    return packed


def conv_bf16_pack_both(weight, h, w):
    """Forward and adjoint packings of a stride-1 layer's weight from ONE launch -> (packed, packed_adjoint)."""
    require_device(weight, "weight")
    wt = weight.contiguous()
    cout, cin, k, _ = wt.shape
    n, na = (lib().sis_conv_bf16_packed_elems(cin, cout, h, w, k, 1, a) for a in (0, 1))
    if wt.dtype not in (torch.float32, torch.bfloat16) or n < 0 or na < 0:
        raise RuntimeError(f"conv_bf16_pack_both: unsupported layer {cin}->{cout} k{k} / dtype {wt.dtype}")
    both = torch.empty(n + na, dtype=torch.bfloat16, device=wt.device)
    packed, adjoint = both[:n], both[n:]
    with torch.cuda.device(wt.device):
        _check(lib().sis_conv_bf16_pack_both(_ptr(packed), _ptr(adjoint), _ptr(wt), _DTYPE_CODE[wt.dtype], cin, cout, h, w, k,
                                             _stream()), "sis_conv_bf16_pack_both")
    return packed, adjoint


def conv_bf16(x, packed, cout, ksize, stride=1, bias=None):
    """x [B,Cin,H,W] bf16 (contiguous), packed weights of ``conv_bf16_pack`` -> [B,cout,Ho,Wo] bf16; padding ksize // 2."""
    require_device(x, "input")
    if x.dtype != torch.bfloat16 or not x.is_contiguous():
        raise RuntimeError("conv_bf16: input must be a contiguous bfloat16 tensor")
    b, cin, h, w = x.shape
    pad = ksize // 2
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    y = torch.empty((b, cout, ho, wo), dtype=torch.bfloat16, device=x.device)
    if bias is not None:
        bias = _f32(bias, "bias")
    with torch.cuda.device(x.device):
        _check(_launch(None, 2.0 * b * cout * cin * ksize * ksize * ho * wo, 2.0 * (x.numel() + y.numel() + packed.numel()),
                       lambda: lib().sis_conv_bf16(_ptr(y), _ptr(x), _ptr(packed), _ptr(bias), b, cin, cout, h, w, ksize,
                                                   stride, _stream())), "sis_conv_bf16")
    return y


def conv_bf16_wgrad_supported(batch, cin, cout, h, w):
    return bool(lib().sis_conv_bf16_wgrad_supported(batch, cin, cout, h, w, WORKSPACE_BYTES))


def conv_bf16_wgrad(x, grad_output, out_dtype=torch.float32, for_param=None, defer=False):
    """dL/dw [Cout,Cin,3,3] (float32 or bfloat16) of a stride-1, padding-1 3x3 convolution from its bf16 input and dL/dy.
    ``defer``: inside a backward the launch may wait for ``flush_deferred`` (batched with the other layers of its shape); the
    caller then guarantees that nobody reads the result before the flush."""
    require_device(x, "input")
    if x.dtype != torch.bfloat16 or grad_output.dtype != torch.bfloat16 or not x.is_contiguous() or not grad_output.is_contiguous():
        raise RuntimeError("conv_bf16_wgrad: input and grad_output must be contiguous bfloat16 tensors")
    b, cin, h, w = x.shape
    cout = grad_output.shape[1]
    dw = grad_out(for_param, (cout, cin, 3, 3), out_dtype, x.device)
    if defer and _conv_wgrad_deferrable():
        _defer_conv_wgrad(3, x, grad_output, dw, (b, cin, cout, h, w))
        return dw
    ws = _workspace(x.device)
    with torch.cuda.device(x.device):
        _check(_launch(None, 2.0 * b * cout * cin * 9 * h * w, 2.0 * (x.numel() + grad_output.numel()) + 4.0 * dw.numel(),
                       lambda: lib().sis_conv_bf16_wgrad(_ptr(dw), _DTYPE_CODE[out_dtype], _ptr(x), _ptr(grad_output), b, cin, cout,
                                                         h, w, _ptr(ws), ws.numel(), _stream())), "sis_conv_bf16_wgrad")
    return dw


def stem_conv_supported(x, weight, stride, padding):
    """The 7x7 stride-2 padding-3 convolution of a 3-channel image to 64 channels (ResNetV2's root) on csrc/stem_conv.hip."""
    return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and weight.dim() == 4
            and weight.dtype in (torch.float32, torch.bfloat16) and weight.shape[2] == weight.shape[3]
            and bool(lib().sis_stem_conv_supported(x.shape[1], weight.shape[0], weight.shape[2], int(stride), int(padding), x.shape[2], x.shape[3])))


def stem_conv_fwd(x, weight):
    """x [B,3,H,W] (float32 or bfloat16), weight [64,3,7,7] -> y [B,64,Ho,Wo] bfloat16 (bf16 products, fp32 sums)."""
    require_device(x, "input")
    x, w = x.contiguous(), weight.contiguous()
    b, _, h, wd = x.shape
    packed = torch.empty(lib().sis_stem_conv_packed_elems(), dtype=torch.bfloat16, device=x.device)
    y = torch.empty((b, 64, (h - 1) // 2 + 1, (wd - 1) // 2 + 1), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().sis_stem_conv_pack(_ptr(packed), _ptr(w), _DTYPE_CODE[w.dtype], _stream()), "sis_stem_conv_pack")
        _check(_launch(None, 2.0 * y.numel() * 147, 2.0 * y.numel() + x.numel() * x.element_size(),
                       lambda: lib().sis_stem_conv_fwd(_ptr(y), _ptr(x), _DTYPE_CODE[x.dtype], _ptr(packed), b, h, wd, _stream())), "sis_stem_conv_fwd")
    return y


def stem_conv_wgrad(x, grad_y, out_dtype=torch.float32, for_param=None):
    """dL/dw [64,3,7,7] of ``stem_conv_fwd`` from the image and dL/dy (bfloat16)."""
    require_device(x, "input")
    x = x.contiguous()
    gy = grad_y.contiguous()
    if gy.dtype != torch.bfloat16:
        gy = gy.bfloat16()
    b, _, h, wd = x.shape
    dw = grad_out(for_param, (64, 3, 7, 7), out_dtype, x.device)
    ws = _workspace(x.device)
    if lib().sis_stem_conv_wgrad_workspace_bytes(b, h, wd) > ws.numel():
        raise RuntimeError("stem_conv_wgrad: split-K workspace too small (SIS_WORKSPACE_MB)")
    with torch.cuda.device(x.device):
        _check(_launch(None, 2.0 * gy.numel() * 147, 2.0 * gy.numel() + x.numel() * x.element_size(),
                       lambda: lib().sis_stem_conv_wgrad(_ptr(dw), _DTYPE_CODE[out_dtype], _ptr(x), _DTYPE_CODE[x.dtype], _ptr(gy), b, h, wd,
                                                         _ptr(ws), ws.numel(), _stream())), "sis_stem_conv_wgrad")
    return dw


class WeightStdPackBank:
    """Standardised + packed bf16 weights of a list of float32 convolution weights, refreshed by ONE launch
    (``sis_weight_std_pack_multi``).  Output buffers are allocated once: ``w_hat[i]``, ``invstd[i]``, ``packed[i]``,
    ``adjoint[i]`` (None for layers without an adjoint image) keep their addresses, ``refresh()`` rewrites their contents."""

    def __init__(self, weights, strides, eps, standardize=True):
        """``standardize=False``: plain layers (``nn.Conv2d`` of the decoder) -- the launch only rounds to bf16 and packs; ``w_hat``
        / ``invstd`` are lists of None."""
        self.eps = float(eps)
        self.standardize = bool(standardize)
        self.weights = list(weights)
        dev = self.weights[0].device
        self.w_hat, self.invstd, self.packed, self.adjoint = [], [], [], []
        rows, row_begin, filter_begin = [], 0, 0
        for w, stride in zip(self.weights, strides):
            cout, cin, k, _ = w.shape
            mt, kc, mt2 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            pe, ae = ctypes.c_int64(), ctypes.c_int64()
            if w.dtype != torch.float32 or not w.is_contiguous() or not lib().sis_weight_std_pack_plan(
                    cin, cout, k, stride, ctypes.byref(mt), ctypes.byref(kc), ctypes.byref(mt2), ctypes.byref(pe), ctypes.byref(ae)):
                raise RuntimeError(f"WeightStdPackBank: no plan for a {cin}->{cout} k{k} s{stride} layer")
            self.w_hat.append(torch.empty(w.shape, dtype=torch.bfloat16, device=dev) if standardize else None)
            self.invstd.append(torch.empty(cout, dtype=torch.float32, device=dev) if standardize else None)
            self.packed.append(torch.empty(pe.value, dtype=torch.bfloat16, device=dev))
            self.adjoint.append(torch.empty(ae.value, dtype=torch.bfloat16, device=dev) if mt2.value else None)
            n_rows = -(-cout // mt.value) * mt.value
            rows.append([w.data_ptr(), self.w_hat[-1].data_ptr() if standardize else 0, self.invstd[-1].data_ptr() if standardize else 0,
                         self.packed[-1].data_ptr(),
                         self.adjoint[-1].data_ptr() if mt2.value else 0, cout, cin, k, mt.value, kc.value, mt2.value, n_rows, row_begin,
                         filter_begin])
            row_begin += n_rows
            filter_begin += cout
        self.total_rows = row_begin
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.source_ptrs = [w.data_ptr() for w in self.weights]

    @staticmethod
    def supported(weight, stride):
        cout, cin, k, k2 = weight.shape
        out = [ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64()]
        return bool(k == k2 and weight.dtype == torch.float32 and weight.is_contiguous()
                    and lib().sis_weight_std_pack_plan(cin, cout, k, stride, *[ctypes.byref(o) for o in out]))

    def current(self):
        """The table still points at the weights (a module moved / re-materialised since would have new storage)."""
        return all(w.data_ptr() == ptr for w, ptr in zip(self.weights, self.source_ptrs))

    def backward(self, grads):
        """grads[i]: dL/dw_hat of layer i or None -> list of dL/dw (float32) or None, one launch for all layers (bf16
        contiguous gradients; anything else goes layer by layer)."""
        n = len(self.weights)
        if any(g is not None and (g.dtype != torch.bfloat16 or not g.is_contiguous()) for g in grads):
            return [None if g is None else weight_std_bwd(g, w, i, self.eps) for g, w, i in zip(grads, self.weights, self.invstd)]
        out = [None if g is None else grad_out(w.data_ptr(), tuple(w.shape), torch.float32, w.device) for g, w in zip(grads, self.weights)]
        g_ptrs = (ctypes.c_void_p * n)(*[None if g is None else g.data_ptr() for g in grads])
        o_ptrs = (ctypes.c_void_p * n)(*[None if o is None else o.data_ptr() for o in out])
        couts = (ctypes.c_int * n)(*[w.shape[0] for w in self.weights])
        with torch.cuda.device(self.table.device):
            _check(lib().sis_weight_std_bwd_multi(_ptr(self.table), g_ptrs, o_ptrs, couts, n, _stream()), "sis_weight_std_bwd_multi")
        return out

    def refresh(self):
        with torch.cuda.device(self.table.device):
            _check(lib().sis_weight_std_pack_multi(_ptr(self.table), len(self.weights), self.total_rows, self.eps, _stream()),
                   "sis_weight_std_pack_multi")


class _SwapLast2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _swap_last2(x)

    @staticmethod
    def backward(ctx, grad):
        return _swap_last2(grad.contiguous())


def _swap_last2(x):
    b, r, c = x.shape
    out = torch.empty((b, c, r), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().sis_transpose_batched(_ptr(out), _ptr(x), x.element_size(), b, r, c, _stream()), "sis_transpose_batched")
    return out


def swap_last2(x):
    """Contiguous [B, R, C] -> contiguous [B, C, R] (differentiable): ``x.transpose(-1, -2).contiguous()`` as one tiled kernel."""
    require_device(x, "input")
    if x.dim() != 3 or x.element_size() not in (2, 4) or not x.is_contiguous():
        raise RuntimeError("swap_last2: a contiguous 3-d tensor of 2- or 4-byte elements is required")
    return _SwapLast2.apply(x)


class TransposeBank:
    """Transposed bf16 copies of a list of 2-D bf16 matrices, refreshed by ONE launch (``sis_transpose_bf16_multi``).
    ``out[i]`` ([cols, rows]) keep their addresses; ``refresh()`` rewrites their contents from the sources."""

    def __init__(self, sources):
        self.sources = list(sources)
        dev = self.sources[0].device
        self.out, rows, first = [], [], 0
        for s in self.sources:
            if s.dtype != torch.bfloat16 or s.dim() != 2 or not s.is_contiguous():
                raise RuntimeError("TransposeBank: sources must be contiguous 2-D bfloat16 tensors")
            r, c = s.shape
            self.out.append(torch.empty((c, r), dtype=torch.bfloat16, device=dev))
            rows.append([s.data_ptr(), self.out[-1].data_ptr(), r, c, first])
            first += -(-r // 64) * -(-c // 64)
        self.total_tiles = first
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.source_ptrs = [s.data_ptr() for s in self.sources]

    def current(self, sources):
        return len(sources) == len(self.source_ptrs) and all(s.data_ptr() == p for s, p in zip(sources, self.source_ptrs))

    def refresh(self):
        with torch.cuda.device(self.table.device):
            _check(lib().sis_transpose_bf16_multi(_ptr(self.table), len(self.sources), self.total_tiles, _stream()),
                   "sis_transpose_bf16_multi")


def conv1x1_bf16_wgrad_supported(batch, cin, cout, pixels):
    return bool(lib().sis_conv1x1_bf16_wgrad_supported(batch, cin, cout, pixels, WORKSPACE_BYTES))


def conv1x1_bf16_wgrad(x, grad_output, out_dtype=torch.float32, for_param=None, defer=False):
    """dL/dw [Cout,Cin,1,1] (float32 or bfloat16) of a stride-1 1x1 convolution from its bf16 NCHW input and dL/dy."""
    require_device(x, "input")
    if x.dtype != torch.bfloat16 or grad_output.dtype != torch.bfloat16 or not x.is_contiguous() or not grad_output.is_contiguous():
        raise RuntimeError("conv1x1_bf16_wgrad: input and grad_output must be contiguous bfloat16 tensors")
    b, cin = x.shape[0], x.shape[1]
    cout = grad_output.shape[1]
    pixels = x[0, 0].numel()
    if grad_output.shape[0] != b or grad_output[0, 0].numel() != pixels:
        raise RuntimeError("conv1x1_bf16_wgrad: input and grad_output disagree on batch / plane size")
    dw = grad_out(for_param, (cout, cin, 1, 1), out_dtype, x.device)
    if defer and _conv_wgrad_deferrable():   # (as conv_bf16_wgrad)
        _defer_conv_wgrad(1, x, grad_output, dw, (b, cin, cout, pixels))
        return dw
    ws = _workspace(x.device)
    with torch.cuda.device(x.device):
        _check(_launch(None, 2.0 * b * cout * cin * pixels, 2.0 * (x.numel() + grad_output.numel()) + 4.0 * dw.numel(),
                       lambda: lib().sis_conv1x1_bf16_wgrad(_ptr(dw), _DTYPE_CODE[out_dtype], _ptr(x), _ptr(grad_output), b, cin, cout,
                                                            pixels, _ptr(ws), ws.numel(), _stream())), "sis_conv1x1_bf16_wgrad")
    return dw


# ------------------------------------------------------------------------------ bf16 GEMM with fused epilogues (ViT encoder)

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
TILE_256X96, TILE_256X192, TILE_256X288 = 9, 10, 11   # 256-row tiles of csrc/gemm256_bf16.hip (NT layout, no split-K)


def gemm_tile_256(m, n, k):
    """Tile code of the 256-row GEMM kernel for an [m, n] output contracted over k, or None when the 128-wide tiles of
    csrc/gemm_bf16.hip suit the shape better: the widest of 256 x 288 / 192 / 96 that divides n and still leaves at least
    ~200 tiles for the 256 compute units (8 192 tokens: n = 2304 -> 256 tiles of 256 x 288, 3072 -> 512 of 256 x 192,
    768 -> 256 of 256 x 96)."""
    if k % 64 or k < 128 or n % 4:
        return None
    m_tiles = -(-m // 256)
    for code, width in ((TILE_256X288, 288), (TILE_256X192, 192), (TILE_256X96, 96)):
        if width in _GEMM256_WIDTHS and n % width == 0 and m_tiles * (n // width) >= 200:
            return code
    return None


# Widths of the 256-row tiles the encoder may pick (tools/bench_gemm256.py, 8 192 tokens, against the 128-wide tiles with the
# same epilogues): 256 x 288 wins on the fused projection (46.2 -> 35.0 us), 256 x 192 on fc1 with its GELU + dropout
# epilogue (72.8 -> 66.3 us); 256 x 96 loses on the 768-wide outputs (its 58 B/clk of operand traffic per MFMA cycle is the
# load path's limit, and a single workgroup per CU cannot hide its epilogue: 47 -> 56 us for fc2).
# SIS_GEMM256_TILES=288,192,96 enables all three for measurements.
_GEMM256_WIDTHS = tuple(int(w) for w in os.environ.get("SIS_GEMM256_TILES", "288,192").split(",") if w)
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU_DROP, EPI_BIAS_DROP_RESID, EPI_GELU_BWD, EPI_F32 = range(6)


def _rows2d(t, name):
    if t.dim() != 2 or t.dtype != torch.bfloat16 or t.stride(1) != 1 or t.stride(0) % 8 or t.data_ptr() % 16:
        raise RuntimeError(f"gemm_bf16: {name} must be a 2-D bfloat16 tensor with unit column stride, a row stride that is a "
                           "multiple of 8 and a 16-byte aligned base")
    return t


def gemm_bf16(a, b, layout, epilogue=EPI_NONE, bias=None, resid=None, pre=None, seed=None, site=0, drop_p=0.0, splits=1, tile=0, out=None):
    """C = epilogue(op(a) op(b)) on the bf16 matrix cores (csrc/gemm_bf16.hip), fp32 accumulation.
      GEMM_NT  a [m,k], b [n,k]   (forward of a Linear layer: x, weight)
      GEMM_NN  a [m,k], b [k,n]   (data gradient: grad, weight)
      GEMM_TN  a [k,m], b [k,n]   (weight gradient: grad, x; EPI_F32, optionally split over k)
    Row-strided views are taken as they are (row stride % 8 == 0).  Returns C (bf16, or fp32 for EPI_BIAS_DROP_RESID /
    EPI_F32); EPI_BIAS_GELU_DROP returns (dropout(gelu(h)), gelu'(h) * dropout factor) -- the second is the ``pre=`` of EPI_GELU_BWD."""
    require_device(a, "a")
    a, b = _rows2d(a, "a"), _rows2d(b, "b")
    if layout == GEMM_NT:
        (m, k), (n, k2) = a.shape, b.shape
    elif layout == GEMM_NN:
        (m, k), (k2, n) = a.shape, b.shape
    else:
        (k, m), (k2, n) = a.shape, b.shape
    if k != k2:
        raise RuntimeError(f"gemm_bf16: contraction lengths differ ({k} vs {k2})")
    f32_out = epilogue in (EPI_BIAS_DROP_RESID, EPI_F32)
    if out is not None:   # (a weight gradient written into its parameter's arena slice: ``grad_out``)
        if tuple(out.shape) != (m, n) or out.dtype != (torch.float32 if f32_out else torch.bfloat16) or not out.is_contiguous():
            raise RuntimeError("gemm_bf16: out must be a contiguous [m, n] tensor of the result's dtype")
        c = out
    else:
        c = torch.empty((m, n), dtype=torch.float32 if f32_out else torch.bfloat16, device=a.device)
    c2 = torch.empty((m, n), dtype=torch.bfloat16, device=a.device) if epilogue == EPI_BIAS_GELU_DROP else None
    b0, b1, b2, seg = bias, None, None, 0
    if isinstance(bias, (tuple, list)):   # query | key | value: three parameters, one fused projection
        b0, b1, b2 = (_f32(t, "bias") for t in bias)
        seg = b0.numel()
    elif bias is not None:
        b0 = _f32(bias, "bias")
    if resid is not None:
        resid = _f32(resid, "residual")
        if tuple(resid.shape) != (m, n):
            raise RuntimeError("gemm_bf16: the residual must have the output's shape")
    if pre is not None and (pre.dtype != torch.bfloat16 or tuple(pre.shape) != (m, n) or not pre.is_contiguous()):
        raise RuntimeError("gemm_bf16: the pre-activation must be a contiguous bfloat16 tensor of the output's shape")
    ws, ws_bytes = None, 0
    if splits > 1:
        # split-K slabs: one scratch buffer per (device, stream) -- the encoder's weight gradients run on a side stream while
        # the main stream's convolution weight gradients use the per-device buffer
        ws = _stream_workspace(a.device)
        ws_bytes = ws.numel()
    name = f"gemm_bf16<{('NT', 'NN', 'TN')[layout]},{epilogue}>" if tile < TILE_256X96 else f"gemm256<{(96, 192, 288)[tile - TILE_256X96]},{epilogue}>"
    with torch.cuda.device(a.device):
        _check(_launch(name, 2.0 * m * n * k, 2.0 * (m * k + n * k) + c.element_size() * m * n,
                       lambda: lib().sis_gemm_bf16(_ptr(c), _ptr(c2), _ptr(a), _ptr(b), layout, epilogue, m, n, k, a.stride(0),
                                                   b.stride(0), n, _ptr(b0), _ptr(b1), _ptr(b2), seg, _ptr(resid), _ptr(pre), _ptr(seed), int(site),
                                                   float(drop_p), int(splits), _ptr(ws), ws_bytes, int(tile), _stream())),
               "sis_gemm_bf16")
    return (c, c2) if c2 is not None else c


def _stream_workspace(device):
    """Split-K scratch of the current stream (the per-device buffer on the default stream)."""
    stream = torch.cuda.current_stream(device)
    if stream == torch.cuda.default_stream(device):
        return _workspace(device)
    ws = _workspaces.get((device, stream.cuda_stream))
    if ws is None:
        ws = _workspaces[(device, stream.cuda_stream)] = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=device)
    return ws


def gemm_bf16_wgrad_bias(grad, x, splits, tile=0, dw=None):
    """Weight and bias gradient of a Linear layer from dL/dy ``grad`` [tokens, out] and its input ``x`` [tokens, in] (bf16):
    -> (dW [out, in] float32, db [out] float32) in the two launches of the split-K weight-gradient GEMM -- the column sums of
    ``grad`` are computed by extra workgroups of those launches (``sis_gemm_bf16_wgrad_bias``)."""
    require_device(grad, "grad")
    grad, x = _rows2d(grad, "grad"), _rows2d(x, "x")
    (k, m), (k2, n) = grad.shape, x.shape
    if k != k2 or splits < 2 or m % 4:
        raise RuntimeError("gemm_bf16_wgrad_bias: row counts differ, fewer than 2 splits, or an output width that is not a multiple of 4")
    if dw is None:
        dw = torch.empty((m, n), dtype=torch.float32, device=grad.device)
    elif tuple(dw.shape) != (m, n) or dw.dtype != torch.float32 or not dw.is_contiguous():
        raise RuntimeError("gemm_bf16_wgrad_bias: dw must be a contiguous float32 [out, in] tensor")
    db = torch.empty(m, dtype=torch.float32, device=grad.device)
    ws = _stream_workspace(grad.device)
    with torch.cuda.device(grad.device):
        _check(_launch("gemm_bf16<TN,5>", 2.0 * m * n * k, 2.0 * (m * k + n * k) + 4.0 * m * n,
                       lambda: lib().sis_gemm_bf16_wgrad_bias(_ptr(dw), _ptr(db), _ptr(grad), _ptr(x), m, n, k, grad.stride(0), x.stride(0),
                                                              int(splits), _ptr(ws), ws.numel(), int(tile), _stream())),
               "sis_gemm_bf16_wgrad_bias")
    return dw, db


_WGRAD_MULTI_TILE = int(os.environ.get("SIS_WGRAD_MULTI_TILE", "4"))   # 128 x 128 four-wave tile of the batched weight gradients


def gemm_bf16_wgrad_bias_multi(jobs, tile=None):
    """Weight and bias gradients of several Linear layers of ONE shape from one launch (+ one for the column sums):
    ``jobs`` = [(grad [tokens, out] bf16, x [tokens, in] bf16, dw [out, in] float32, db [out] float32), ...] -- or the same as raw
    addresses (grad_ptr, x_ptr, dw_ptr, db_ptr) with ``shape`` = (tokens, out, in, lda, ldb, device) given through ``tile``'s
    sibling ``_wgrad_multi_raw`` (the deferred queue).  Every problem contracts all its tokens per tile: no split-K slabs."""
    g0, x0 = jobs[0][0], jobs[0][1]
    require_device(g0, "grad")
    k, m = g0.shape
    n = x0.shape[1]
    for g, x, dw, db in jobs:
        if (tuple(g.shape) != (k, m) or tuple(x.shape) != (k, n) or g.dtype != torch.bfloat16 or x.dtype != torch.bfloat16
                or g.stride(1) != 1 or x.stride(1) != 1 or g.stride(0) != g0.stride(0) or x.stride(0) != x0.stride(0)
                or tuple(dw.shape) != (m, n) or dw.dtype != torch.float32 or not dw.is_contiguous()
                or tuple(db.shape) != (m,) or db.dtype != torch.float32):
            raise RuntimeError("gemm_bf16_wgrad_bias_multi: every job must have the first job's shapes, strides and dtypes")
    _wgrad_multi_raw([(g.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr()) for g, x, dw, db in jobs],
                     (k, m, n, g0.stride(0), x0.stride(0), g0.device), tile)


def _wgrad_multi_raw(ptrs, shape, tile=None):
    k, m, n, lda, ldb, device = shape
    nj = len(ptrs)
    arr = ctypes.c_void_p * nj
    ws = _stream_workspace(device)
    need = lib().sis_gemm_bf16_wgrad_multi_workspace_bytes(min(nj, 16), m, k)
    if need > ws.numel():
        raise RuntimeError("gemm_bf16_wgrad_bias_multi: split-K workspace too small for the partial column sums")
    with torch.cuda.device(device):
        _check(_launch("gemm_bf16<TN,multi>", 2.0 * m * n * k * nj, (2.0 * (m * k + n * k) + 4.0 * m * n) * nj,
                       lambda: lib().sis_gemm_bf16_wgrad_bias_multi(arr(*[p[2] for p in ptrs]), arr(*[p[3] for p in ptrs]),
                                                                    arr(*[p[0] for p in ptrs]), arr(*[p[1] for p in ptrs]), nj, m, n, k,
                                                                    lda, ldb, _ptr(ws), ws.numel(),
                                                                    _WGRAD_MULTI_TILE if tile is None else int(tile), _stream())),
               "sis_gemm_bf16_wgrad_bias_multi")


def _batched_operand(t, name):
    """[batches, rows, cols] (or [rows, cols], shared by every batch entry) -> (tensor, rows, cols, row stride, batch stride)."""
    if t.dim() == 2:
        t = _rows2d(t, name)
        return t, t.shape[0], t.shape[1], t.stride(0), 0, 1
    if t.dim() != 3 or t.dtype != torch.bfloat16 or t.stride(2) != 1 or t.stride(1) % 8 or t.data_ptr() % 16 or t.stride(0) % 8:
        raise RuntimeError(f"gemm_bf16_batched: {name} must be [batches, rows, cols] bfloat16 with unit column stride and 16-byte aligned rows")
    return t, t.shape[1], t.shape[2], t.stride(1), t.stride(0), t.shape[0]


def gemm_bf16_batched(a, b, layout, epilogue=EPI_NONE, sum_over_batches=False, tile=0):
    """``gemm_bf16`` over a batch of problems in one launch; 2-D operands are shared by all entries.  ``sum_over_batches``:
    one fp32 result = the sum of the entries' products (EPI_F32; 1, 2, 4 or a multiple of 8 entries)."""
    require_device(a, "a")
    a, ar, ac, lda, abs_, na = _batched_operand(a, "a")
    b, br, bc, ldb, bbs, nb = _batched_operand(b, "b")
    batches = max(na, nb)
    if (na not in (1, batches)) or (nb not in (1, batches)):
        raise RuntimeError("gemm_bf16_batched: batch counts differ")
    if layout == GEMM_NT:
        (m, k), (n, k2) = (ar, ac), (br, bc)
    elif layout == GEMM_NN:
        (m, k), (k2, n) = (ar, ac), (br, bc)
    else:
        (k, m), (k2, n) = (ar, ac), (br, bc)
    if k != k2:
        raise RuntimeError(f"gemm_bf16_batched: contraction lengths differ ({k} vs {k2})")
    dtype = torch.float32 if epilogue == EPI_F32 else torch.bfloat16
    c = torch.empty((m, n) if sum_over_batches else (batches, m, n), dtype=dtype, device=a.device)
    ws, ws_bytes = None, 0
    if sum_over_batches and batches > 1:
        ws = _workspace(a.device)
        ws_bytes = ws.numel()
    with torch.cuda.device(a.device):
        _check(_launch(f"gemm_bf16_batched<{('NT', 'NN', 'TN')[layout]}>", 2.0 * batches * m * n * k,
                       2.0 * batches * (m * k + n * k) + c.element_size() * c.numel(),
                       lambda: lib().sis_gemm_bf16_batched(_ptr(c), _ptr(a), _ptr(b), layout, epilogue, m, n, k, lda, ldb, n, batches,
                                                           abs_, bbs, 0 if sum_over_batches else m * n, int(bool(sum_over_batches)),
                                                           _ptr(ws), ws_bytes, int(tile), _stream())), "sis_gemm_bf16_batched")
    return c


def attention_fwd(qkv, heads):
    """qkv bf16 [B, N, 3 * heads * 64] (query | key | value) -> (context bf16 [B, N, heads * 64], lse fp32 [B, heads, N])."""
    require_device(qkv, "qkv")
    if qkv.dtype != torch.bfloat16 or qkv.dim() != 3 or not qkv.is_contiguous() or qkv.shape[2] != 3 * heads * 64:
        raise RuntimeError("attention_fwd: qkv must be a contiguous bfloat16 [B, N, 3 * heads * 64] tensor")
    b, n, _ = qkv.shape
    ctx = torch.empty((b, n, heads * 64), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((b, heads, n), dtype=torch.float32, device=qkv.device)
    flops = 4.0 * b * heads * n * n * 64
    with torch.cuda.device(qkv.device):
        _check(_launch("attn_fwd_kernel", flops, 2.0 * (qkv.numel() + ctx.numel()),
                       lambda: lib().sis_attention_fwd(_ptr(ctx), _ptr(lse), _ptr(qkv), b, n, heads, _stream())), "sis_attention_fwd")
    return ctx, lse


def attention_bwd(d_ctx, qkv, ctx, lse, heads):
    """Gradient of ``attention_fwd`` w.r.t. qkv (bf16, qkv's layout)."""
    if d_ctx.dtype != torch.bfloat16 or not d_ctx.is_contiguous() or d_ctx.shape != ctx.shape:
        raise RuntimeError("attention_bwd: d_ctx must be a contiguous bfloat16 tensor of the context's shape")
    b, n, _ = qkv.shape
    d_qkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    flops = 14.0 * b * heads * n * n * 64   # 7 products (S and dP are recomputed by both kernels)
    with torch.cuda.device(qkv.device):
        _check(_launch("attn_bwd_kernels", flops, 2.0 * (2 * qkv.numel() + 2 * ctx.numel()),
                       lambda: lib().sis_attention_bwd(_ptr(d_qkv), _ptr(delta), _ptr(d_ctx), _ptr(qkv), _ptr(ctx), _ptr(lse), b, n,
                                                       heads, _stream())), "sis_attention_bwd")
    return d_qkv


def ce_dice_supported(logits, labels):
    return (logits.is_cuda and logits.dtype in (torch.float32, torch.bfloat16) and logits.dim() >= 3 and 2 <= logits.shape[1] <= 8
            and labels.dtype == torch.int64 and (logits[0, 0].numel() % 4 == 0))


def ce_dice_fwd(logits, labels):
    """logits [B,C,...] (f32 / bf16), labels int64 [B,...] -> (out3 = [0.5 ce + 0.5 dice, ce, dice], stats for the backward)."""
    require_device(logits, "logits")
    logits, labels = logits.contiguous(), labels.contiguous()
    b, c = logits.shape[:2]
    hw = logits[0, 0].numel()
    out = torch.empty(3, dtype=torch.float32, device=logits.device)
    stats = torch.empty(1 + 2 * c, dtype=torch.float32, device=logits.device)
    ws = torch.empty(lib().sis_ce_dice_workspace_floats(c), dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device):
        _check(lib().sis_ce_dice_fwd(_ptr(out), _ptr(stats), _ptr(ws), _ptr(logits), _DTYPE_CODE[logits.dtype], _ptr(labels), b, c, hw,
                                     _stream()), "sis_ce_dice_fwd")
    return out, stats


def ce_dice_bwd(grad_loss, logits, labels, stats):
    logits, labels = logits.contiguous(), labels.contiguous()
    b, c = logits.shape[:2]
    hw = logits[0, 0].numel()
    grad = torch.empty_like(logits)
    if grad_loss is not None:
        grad_loss = _f32(grad_loss, "grad_loss").reshape(1)
    with torch.cuda.device(logits.device):
        _check(lib().sis_ce_dice_bwd(_ptr(grad), _ptr(logits), _DTYPE_CODE[logits.dtype], _ptr(labels), _ptr(stats), _ptr(grad_loss), b, c,
                                     hw, _stream()), "sis_ce_dice_bwd")
    return grad


_drop_seeds = {}


def dropout_seed(device):
    """The device seed word (int64 [1]) every dropout site of the fused ViT kernels reads; initialised from torch's seed."""
    device = torch.device(device)
    t = _drop_seeds.get(device)
    if t is None:
        t = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)
        _drop_seeds[device] = t
    return t


def dropout_advance(seed):
    with torch.cuda.device(seed.device):
        _check(lib().sis_dropout_advance(_ptr(seed), _stream()), "sis_dropout_advance")


def dropout_bwd_cast(grad, seed, site, drop_p):
    """bf16(grad * dropout factor): gradient of ``resid + dropout(y)`` w.r.t. y (y dense, row-major, grad fp32)."""
    grad = _f32(grad, "grad")
    out = torch.empty(grad.shape, dtype=torch.bfloat16, device=grad.device)
    with torch.cuda.device(grad.device):
        _check(lib().sis_dropout_bwd_cast(_ptr(out), _ptr(grad), grad.numel(), _ptr(seed), int(site), float(drop_p), _stream()),
               "sis_dropout_bwd_cast")
    return out


# ------------------------------------------------------------------------------ layer norm


def layer_norm_supported(x, n):
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and n in (256, 512, 768, 1024) and x.shape[-1] == n


def layer_norm_fwd(x, gamma, beta, eps, out_dtype=None):
    """x [..., n] (f32 / bf16) -> (y in ``out_dtype``, mean [rows], rstd [rows])."""
    require_device(x, "input")
    x = x.contiguous()
    out_dtype = out_dtype or x.dtype
    n = x.shape[-1]
    rows = x.numel() // n
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    with torch.cuda.device(x.device):
        _check(lib().sis_layer_norm_fwd(_ptr(y), _ptr(mean), _ptr(rstd), _ptr(x), _ptr(_f32(gamma, "weight")),
                                        _ptr(_f32(beta, "bias")), _DTYPE_CODE[x.dtype], _DTYPE_CODE[out_dtype], rows, n,
                                        float(eps), _stream()), "sis_layer_norm_fwd")
    return y, mean, rstd


def layer_norm_bwd_fused(grad_y, x, mean, rstd, gamma, residual_grad=None, cast_seed=None, cast_site=None, cast_p=0.0, defer=False):
    """LayerNorm backward of a pre-norm residual block: dx = residual_grad + LN'(grad_y); with ``cast_site`` also returns
    bf16(dx * dropout factor of that site) -> (dx, dgamma, dbeta, cast or None).  ``defer``: the caller vouches that the two
    parameters hold NO gradient yet (autograd will then take dgamma / dbeta as ``.grad`` as they are -- with a gradient in place it
    would ADD them on the spot); inside a backward their reduction then joins the deferred queue (``flush_deferred``)."""
    x = x.contiguous()
    g = grad_y.contiguous()
    n = x.shape[-1]
    rows = x.numel() // n
    dx = torch.empty_like(x)
    dgamma = torch.empty(n, dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    cast = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if cast_site is not None else None
    if residual_grad is not None:
        residual_grad = _f32(residual_grad, "residual_grad")
    ws = torch.empty(lib().sis_layer_norm_workspace_floats(n), dtype=torch.float32, device=x.device)
    if defer and deferring():   # d(gamma) / d(beta) by the batched reduction at the end of the backward (flush_deferred)
        with torch.cuda.device(x.device):
            _check(lib().sis_layer_norm_bwd_fused_partial(_ptr(dx), _ptr(ws), _ptr(g), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma),
                                                          _DTYPE_CODE[x.dtype], _DTYPE_CODE[g.dtype], rows, n, _ptr(residual_grad),
                                                          _ptr(cast), _ptr(cast_seed), int(cast_site or 0), float(cast_p), _stream()),
                   "sis_layer_norm_bwd_fused_partial")
        _deferred["ln"].append((dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), int(lib().sis_layer_norm_bwd_parts(rows)), n, x.device,
                                _hold(dgamma, dbeta, ws)))
        return dx, dgamma, dbeta, cast
    with torch.cuda.device(x.device):
        _check(lib().sis_layer_norm_bwd_fused(_ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws), _ptr(g), _ptr(x), _ptr(mean), _ptr(rstd),
                                              _ptr(gamma), _DTYPE_CODE[x.dtype], _DTYPE_CODE[g.dtype], rows, n, _ptr(residual_grad),
                                              _ptr(cast), _ptr(cast_seed), int(cast_site or 0), float(cast_p), _stream()),
               "sis_layer_norm_bwd_fused")
    return dx, dgamma, dbeta, cast


def layer_norm_bwd(grad_y, x, mean, rstd, gamma):
    x = x.contiguous()
    g = grad_y.contiguous()
    if g.dtype not in (torch.float32, torch.bfloat16):
        g = g.float()
    n = x.shape[-1]
    rows = x.numel() // n
    dx = torch.empty_like(x)
    dgamma = torch.empty(n, dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    ws = torch.empty(lib().sis_layer_norm_workspace_floats(n), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().sis_layer_norm_bwd(_ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws), _ptr(g), _ptr(x), _ptr(mean), _ptr(rstd),
                                        _ptr(gamma), _DTYPE_CODE[x.dtype], _DTYPE_CODE[g.dtype], rows, n, _stream()),
               "sis_layer_norm_bwd")
    return dx, dgamma, dbeta


def column_sum(x):
    """x [rows, n] (f32 / f16 / bf16, n % 4 == 0) -> float32 [n] column sums (bias gradient of a Linear layer)."""
    require_device(x, "input")
    x = x.contiguous()
    rows, n = x.shape
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib().sis_column_sum_workspace_floats(n), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().sis_column_sum(_ptr(out), _ptr(ws), _ptr(x), _DTYPE_CODE[x.dtype], rows, n, _stream()), "sis_column_sum")
    return out


# ------------------------------------------------------------------------------ group norm (+ ReLU)


_GROUP_COUNTERS = {}  # (device, stream) -> int32 zeros: completion counters of the norm kernels (each launch leaves them zero)
_GROUP_COUNTER_ROWS = 1 << 16   # sized up front: a captured hipGraph holds the buffer's address, so it is never reallocated
_GN_FUSED_FINISH = os.environ.get("SIS_GN_FUSED_FINISH", "1") != "0"  # 0: the per-group merges as launches of their own


def _group_counters(device, n):
    """Completion counters of the in-launch hand-over (csrc/sis_xwg.h), one buffer per (device, stream) like the split-K
    scratch: launches on one stream are ordered, and every launch leaves its counters at zero, so concurrent streams must not
    share them.  The buffer has a fixed size -- pointers baked into a captured graph stay valid -- and a launch
    that would need more rows falls back to the merge-as-its-own-launch form (returns None)."""
    if not _GN_FUSED_FINISH or n > _GROUP_COUNTER_ROWS:
        return None
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _GROUP_COUNTERS.get(key)
    if buf is None:
        buf = _GROUP_COUNTERS[key] = torch.zeros(_GROUP_COUNTER_ROWS, dtype=torch.int32, device=device)
    return buf


def prepare_capture(device):
    """Called on the stream a hipGraph capture is about to run on, BEFORE the capture opens: creates (and zeroes, as ordinary
    eager stream work) the per-stream state that captured launches will hold pointers to -- the completion counters and the
    split-K scratch -- so that neither lives in the graph's private pool with its zero-fill as a captured node only (ADVICE r4:
    a failed capture would leave a cached, never-zeroed counter buffer behind for the next capture on that stream).
    Returns a snapshot for ``rollback_capture``."""
    _group_counters(device, 1)
    _stream_workspace(device)
    return (set(_GROUP_COUNTERS), set(_workspaces))


def rollback_capture(snapshot):
    """After a FAILED capture: forget per-stream buffers that were first created while the capture was open."""
    counters, workspaces = snapshot
    for key in [k for k in _GROUP_COUNTERS if k not in counters]:
        del _GROUP_COUNTERS[key]
    for key in [k for k in _workspaces if k not in workspaces]:
        del _workspaces[key]


def group_counters_are_zero():
    """Debug check (tests): every completion-counter buffer is back at zero, i.e. no launch left a hand-over half done."""
    return all(int(buf.abs().max().item()) == 0 for buf in _GROUP_COUNTERS.values())


def group_norm_fwd(x, gamma, beta, groups, eps, relu, out_dtype=None, residual=None, low_precision_copy=False, want_gate=False):
    """x [B,C,...] (f32 / f16 / bf16) -> (y in ``out_dtype`` (default: x's), mean [B*groups], rstd [B*groups]).
    ``residual`` (float32, x's shape) is added before the ReLU; the output is then float32.
    ``low_precision_copy`` (16-bit x, float32 y): also returns y rounded to x's dtype, written in the same pass
    -> (y, mean, rstd, y_lp).  ``want_gate`` (with relu): one more result, the ReLU gate [y > 0] as one bit per element (uint8
    tensor) that ``group_norm_bwd(..., gate=...)`` reads instead of the saved output."""
    require_device(x, "input")
    x = x.contiguous()
    if residual is not None:
        out_dtype = torch.float32
        residual = _f32(residual, "residual")
        if residual.shape != x.shape:
            raise RuntimeError("residual must have the input's shape")
    out_dtype = out_dtype or x.dtype
    if low_precision_copy and not (out_dtype == torch.float32 and x.dtype in (torch.float16, torch.bfloat16)):
        raise RuntimeError("low_precision_copy needs a float32 output of a 16-bit input")
    b, c = x.shape[0], x.shape[1]
    hw = x[0, 0].numel()
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    y_lp = torch.empty_like(x) if low_precision_copy else None
    mean = torch.empty(b * groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    ws = torch.empty(lib().sis_group_norm_workspace_floats(b, c, hw), dtype=torch.float32, device=x.device)
    gate = torch.empty(lib().sis_group_norm_gate_bytes(b, c, hw), dtype=torch.uint8, device=x.device) if (want_gate and relu) else None
    with torch.cuda.device(x.device):
        _check(lib().sis_group_norm_fwd(_ptr(y), _ptr(y_lp), _ptr(mean), _ptr(rstd), _ptr(ws), _ptr(x), _ptr(residual),
                                        _ptr(_f32(gamma, "weight")), _ptr(_f32(beta, "bias")), _DTYPE_CODE[x.dtype],
                                        _DTYPE_CODE[out_dtype], b, c, hw, groups, float(eps), int(bool(relu)),
                                        _ptr(_group_counters(x.device, b * groups)), _ptr(gate), _stream()),
               "sis_group_norm_fwd")
    out = (y, mean, rstd, y_lp) if low_precision_copy else (y, mean, rstd)
    return out + (gate,) if want_gate else out


def group_norm_bwd(grad_y, x, mean, rstd, gamma, beta, groups, relu, y_mask=None, want_residual_grad=False, grad_y_lp=None, gate=None):
    """-> (dx, dgamma, dbeta[, dresidual]).  ``y_mask``: the saved float32 output when a residual was added (its sign is
    the ReLU mask); ``want_residual_grad`` also returns the gradient of the residual branch; ``grad_y_lp`` (x's 16-bit
    dtype): the gradient that came back through the low-precision copy of the output, added to ``grad_y`` (float32) on load."""
    x = x.contiguous()
    g = grad_y.contiguous()
    if g.dtype != x.dtype and g.dtype != torch.float32:
        g = g.to(x.dtype)
    if grad_y_lp is not None:
        if g.dtype != torch.float32 or grad_y_lp.dtype != x.dtype or x.dtype == torch.float32:
            raise RuntimeError("grad_y_lp needs a float32 grad_y and the 16-bit dtype of x")
        grad_y_lp = grad_y_lp.contiguous()
    b, c = x.shape[0], x.shape[1]
    hw = x[0, 0].numel()
    dx = torch.empty_like(x)
    dres = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_residual_grad else None
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    ws = torch.empty(lib().sis_group_norm_workspace_floats(b, c, hw), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().sis_group_norm_bwd(_ptr(dx), _ptr(dres), _ptr(dgamma), _ptr(dbeta), _ptr(ws), _ptr(g), _ptr(grad_y_lp), _ptr(x),
                                        _ptr(y_mask), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _DTYPE_CODE[x.dtype],
                                        _DTYPE_CODE[g.dtype], b, c, hw, groups, int(bool(relu)),
                                        _ptr(_group_counters(x.device, b * groups)), _ptr(gate), _stream()), "sis_group_norm_bwd")
    return (dx, dgamma, dbeta, dres) if want_residual_grad else (dx, dgamma, dbeta)


def batch_norm_train_fwd(x, gamma, beta, running_mean, running_var, eps, momentum, relu, out_dtype=None):
    """Training-mode batch norm (+ ReLU) on f32 / f16 / bf16 tensors -> (y, mean [C], rstd [C])."""
    require_device(x, "input")
    x = x.contiguous()
    out_dtype = out_dtype or x.dtype
    b, c = x.shape[0], x.shape[1]
    hw = x[0, 0].numel()
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    ws = torch.empty(lib().sis_group_norm_workspace_floats(b, c, hw), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().sis_batch_norm_fwd(_ptr(y), _ptr(mean), _ptr(rstd), _ptr(running_mean), _ptr(running_var), _ptr(ws), _ptr(x),
                                        _ptr(_f32(gamma, "weight")), _ptr(_f32(beta, "bias")), _DTYPE_CODE[x.dtype],
                                        _DTYPE_CODE[out_dtype], b, c, hw, float(eps), float(momentum), int(bool(relu)),
                                        _ptr(_group_counters(x.device, c)), _stream()), "sis_batch_norm_fwd")
    return y, mean, rstd


def batch_norm_train_bwd(grad_y, x, mean, rstd, gamma, beta, relu):
    x = x.contiguous()
    g = grad_y.contiguous()
    if g.dtype != x.dtype and g.dtype != torch.float32:
        g = g.to(x.dtype)
    b, c = x.shape[0], x.shape[1]
    hw = x[0, 0].numel()
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    ws = torch.empty(lib().sis_group_norm_workspace_floats(b, c, hw), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().sis_batch_norm_bwd(_ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws), _ptr(g), _ptr(x), _ptr(mean), _ptr(rstd),
                                        _ptr(gamma), _ptr(beta), _DTYPE_CODE[x.dtype], _DTYPE_CODE[g.dtype], b, c, hw,
                                        int(bool(relu)), _ptr(_group_counters(x.device, c)), _stream()), "sis_batch_norm_bwd")
    return dx, dgamma, dbeta


# ------------------------------------------------------------------------------ weight standardisation


def weight_std_fwd(weight, eps, out_dtype=torch.float32):
    """weight [Cout, ...] float32 -> (w_hat in ``out_dtype``, invstd [Cout]) standardised per output channel."""
    w = _f32(weight, "weight")
    rows, n = w.shape[0], w[0].numel()
    w_hat = torch.empty(w.shape, dtype=out_dtype, device=w.device)
    invstd = torch.empty(rows, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _check(lib().sis_weight_std_fwd(_ptr(w_hat), _ptr(invstd), _ptr(w), _DTYPE_CODE[out_dtype], rows, n, float(eps),
                                        _stream()), "sis_weight_std_fwd")
    return w_hat, invstd


def weight_std_bwd(grad_w_hat, weight, invstd, eps):
    w = _f32(weight, "weight")
    g = grad_w_hat.contiguous()
    rows, n = w.shape[0], w[0].numel()
    dw = torch.empty_like(w)
    with torch.cuda.device(w.device):
        _check(lib().sis_weight_std_bwd(_ptr(dw), _ptr(g), _ptr(w), _ptr(invstd), _DTYPE_CODE[g.dtype], rows, n, float(eps),
                                        _stream()), "sis_weight_std_bwd")
    return dw


# ------------------------------------------------------------------------------ bilinear upsampling (align_corners)


def upsample_bilinear(x, out_h, out_w, grad_output=None):
    """Forward: x [B,C,H,W] -> [B,C,out_h,out_w].  With ``grad_output`` [B,C,out_h,out_w]: the gradient w.r.t. an input
    of x's shape (x is only used for its shape / dtype)."""
    require_device(x, "input")
    if x.dtype not in _DTYPE_CODE or x.dtype == torch.float64:
        raise RuntimeError(f"upsample_bilinear: dtype {x.dtype} not supported")
    b, c, h, w = x.shape
    if grad_output is None:
        src, out, backward = x.contiguous(), torch.empty((b, c, out_h, out_w), dtype=x.dtype, device=x.device), 0
    else:
        src, out, backward = grad_output.contiguous(), torch.empty((b, c, h, w), dtype=x.dtype, device=x.device), 1
    with torch.cuda.device(x.device):
        _check(lib().sis_upsample_bilinear(_ptr(out), _ptr(src), _DTYPE_CODE[x.dtype], b * c, h, w, out_h, out_w, backward,
                                           _stream()), "sis_upsample_bilinear")
    return out


def upsample2x_into(out, x):
    """Bilinear x2 upsampling (align_corners=True) of x [B,C,H,W] into the leading C channels of ``out`` [B,C+S,2H,2W]
    (contiguous; the remaining S channels are left alone)."""
    require_device(x, "input")
    b, c, h, w = x.shape
    if out.shape[0] != b or out.shape[1] < c or out.shape[2:] != (2 * h, 2 * w) or out.dtype != x.dtype or not out.is_contiguous():
        raise RuntimeError("upsample2x_into: out must be a contiguous [B, >= C, 2H, 2W] tensor of the input's dtype")
    with torch.cuda.device(x.device):
        _check(lib().sis_upsample_bilinear_strided(_ptr(out), _ptr(x.contiguous()), _DTYPE_CODE[x.dtype], b, c, h, w, 2 * h, 2 * w,
                                                   out.stride(0), 0, _stream()), "sis_upsample_bilinear_strided")
    return out


def upsample2x_grad_from(grad_wide, channels):
    """Gradient of ``upsample2x_into`` w.r.t. x from the gradient of the WIDE tensor [B,C+S,2H,2W] (contiguous; 16-byte aligned
    images): reads its leading ``channels`` channels in place -> [B,channels,H,W]."""
    require_device(grad_wide, "grad")
    b, ct, oh, ow = grad_wide.shape
    if not grad_wide.is_contiguous() or oh % 2 or ow % 2 or channels > ct:
        raise RuntimeError("upsample2x_grad_from: a contiguous [B, >= C, 2H, 2W] gradient is required")
    gx = torch.empty((b, channels, oh // 2, ow // 2), dtype=grad_wide.dtype, device=grad_wide.device)
    with torch.cuda.device(grad_wide.device):
        _check(lib().sis_upsample_bilinear_strided(_ptr(gx), _ptr(grad_wide), _DTYPE_CODE[grad_wide.dtype], b, channels, oh // 2, ow // 2,
                                                   oh, ow, grad_wide.stride(0), 1, _stream()), "sis_upsample_bilinear_strided")
    return gx


def conv1x1_wgrad_f32_supported(grad_output, input):
    return bool(grad_output.is_cuda and input.is_cuda and grad_output.dtype == torch.float32 and input.dtype == torch.float32
                and grad_output.dim() == 4 and input.dim() == 4 and grad_output.is_contiguous() and input.is_contiguous()
                and lib().sis_conv1x1_wgrad_f32_supported(input.shape[0], input.shape[1], grad_output.shape[1],
                                                         input.shape[2] * input.shape[3]))


def conv1x1_wgrad_f32(grad_output, input, for_param=None, defer=False):
    """dW [Cout, Cin, 1, 1] of a 1x1 stride-1 convolution from dL/dy [B,Cout,H,W] and x [B,Cin,H,W] (fp32, NCHW).  ``defer``: as
    ``conv_bf16_wgrad``."""
    require_device(input, "input")
    b, cin, h, w = input.shape
    cout = grad_output.shape[1]
    L = lib()
    dw = grad_out(for_param, (cout, cin, 1, 1), torch.float32, input.device)
    if (defer and _conv_wgrad_deferrable() and grad_output.dtype == torch.float32 and input.dtype == torch.float32
            and grad_output.is_contiguous() and input.is_contiguous() and grad_output.data_ptr() % 16 == 0 and input.data_ptr() % 16 == 0):
        _defer_conv_wgrad("f1", input, grad_output, dw, (b, cin, cout, h * w))
        return dw
    ws_bytes = int(L.sis_conv1x1_wgrad_f32_workspace(b, cin, cout, h * w))
    ws = torch.empty(max(ws_bytes // 4, 4), dtype=torch.float32, device=input.device)
    with torch.cuda.device(input.device):
        _check(_launch("conv1x1_wgrad_f32_kernel", 2.0 * b * cout * cin * h * w, 4.0 * (grad_output.numel() + input.numel() + dw.numel()),
                       lambda: L.sis_conv1x1_wgrad_f32(_ptr(dw), _ptr(grad_output), _ptr(input), b, cin, cout, h * w,
                                                       _ptr(ws) if ws_bytes else None, ws_bytes, _stream())), "sis_conv1x1_wgrad_f32")
    return dw


# ------------------------------------------------------------------------------ max pooling


def max_pool2d(x, kernel, stride, padding):
    """x [B,C,H,W] -> (out [B,C,Ho,Wo], argmax uint8 [B,C,Ho,Wo]); floor mode, dilation 1 (ATen's output size rule)."""
    require_device(x, "input")
    if x.dtype not in _DTYPE_CODE or x.dtype == torch.float64:
        raise RuntimeError(f"max_pool2d: dtype {x.dtype} not supported")
    b, c, h, w = x.shape
    oh, ow = (h + 2 * padding - kernel) // stride + 1, (w + 2 * padding - kernel) // stride + 1
    x = x.contiguous()
    out = torch.empty((b, c, oh, ow), dtype=x.dtype, device=x.device)
    arg = torch.empty((b, c, oh, ow), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().sis_max_pool2d(_ptr(out), _ptr(arg), _ptr(x), _DTYPE_CODE[x.dtype], b * c, h, w, oh, ow, kernel, stride,
                                    padding, 0, _stream()), "sis_max_pool2d")
    return out, arg


def max_pool2d_backward(grad_output, argmax, in_h, in_w, kernel, stride, padding):
    require_device(grad_output, "grad_output")
    b, c, oh, ow = grad_output.shape
    g = grad_output.contiguous()
    dx = torch.empty((b, c, in_h, in_w), dtype=g.dtype, device=g.device)
    with torch.cuda.device(g.device):
        _check(lib().sis_max_pool2d(_ptr(dx), _ptr(argmax), _ptr(g), _DTYPE_CODE[g.dtype], b * c, in_h, in_w, oh, ow, kernel,
                                    stride, padding, 1, _stream()), "sis_max_pool2d")
    return dx


# ------------------------------------------------------------------------------ patch-wise page inference


def _grid(xs, ys, device):
    xs = torch.as_tensor(xs, dtype=torch.int32).to(device).contiguous()
    ys = torch.as_tensor(ys, dtype=torch.int32).to(device).contiguous()
    return xs, ys


def crop_patches_u8(image, xs, ys, patch):
    """uint8 [H,W,C] page on the device -> float32 [len(ys)*len(xs), C, patch, patch] in [-1,1] (zero padding outside
    the page, ToTensor + Normalize(0.5, 0.5)); patch n = yi * len(xs) + xi."""
    require_device(image, "image")
    if image.dtype != torch.uint8 or image.dim() != 3:
        raise RuntimeError("image must be a uint8 [H, W, C] tensor")
    image = image.contiguous()
    h, w, c = image.shape
    xs, ys = _grid(xs, ys, image.device)
    out = torch.empty((ys.numel() * xs.numel(), c, patch, patch), dtype=torch.float32, device=image.device)
    with torch.cuda.device(image.device):
        _check(lib().sis_crop_patches_u8(_ptr(out), _ptr(image), _ptr(xs), _ptr(ys), xs.numel(), ys.numel(), h, w, c,
                                         patch, _stream()), "sis_crop_patches_u8")
    return out


def assemble_max(pred, xs, ys, height, width, with_labels=False):
    """[N,C,P,P] patch predictions -> [C,height,width] maximum over the covering patches (+ uint8 label map)."""
    pred = _f32(pred, "predictions")
    n, c, p, p2 = pred.shape
    xs, ys = _grid(xs, ys, pred.device)
    if p != p2 or n != xs.numel() * ys.numel():
        raise RuntimeError(f"predictions {tuple(pred.shape)} do not match a {ys.numel()} x {xs.numel()} patch grid")
    out = torch.empty((c, height, width), dtype=torch.float32, device=pred.device)
    labels = torch.empty((height, width), dtype=torch.uint8, device=pred.device) if with_labels else None
    with torch.cuda.device(pred.device):
        _check(lib().sis_assemble_max(_ptr(out), _ptr(labels), _ptr(pred), _ptr(xs), _ptr(ys), xs.numel(), ys.numel(), c,
                                      height, width, p, _stream()), "sis_assemble_max")
    return (out, labels) if with_labels else out


# ------------------------------------------------------------------------------ fused batch norm


def bn_supported(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and (x.shape[2] * x.shape[3]) % 4 == 0 and x.is_contiguous()


def bn_stats(x, running_mean, running_var, eps, momentum):
    b, c, h, w = x.shape
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    invstd = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib().sis_bn_workspace_floats(b, c, h * w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().sis_bn_stats(_ptr(mean), _ptr(invstd), _ptr(running_mean), _ptr(running_var), _ptr(x), _ptr(ws), b, c,
                                  h * w, float(eps), float(momentum), _stream()), "sis_bn_stats")
    return mean, invstd


def bn_fused_supported(x):
    b, c, h, w = x.shape
    return bool(lib().sis_bn_fused_supported(b, c, h * w))


def bn_fused_fwd(x, residual, gamma, beta, running_mean, running_var, eps, momentum, relu, want_mask=False):
    """``bn_stats`` + ``bn_act_fwd`` in one launch (a channel's batch * H * W values fit one workgroup: ``bn_fused_supported``)
    -> (y, mean, invstd, mask or None)."""
    b, c, h, w = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    invstd = torch.empty(c, dtype=torch.float32, device=x.device)
    mask = torch.empty(lib().sis_bn_mask_words(b, c, h * w), dtype=torch.int64, device=x.device) if (want_mask and relu) else None
    with torch.cuda.device(x.device):
        # (kernel name reported by the library: bn_fused_fwd_kernel, or bn_wide_fwd_kernel for channels of 16 385 ... 65 536 values)
        _check(_launch(None, 0.0, 4.0 * (2 + (residual is not None)) * x.numel(), lambda: lib().sis_bn_fused_fwd(
            _ptr(y), _ptr(mean), _ptr(invstd), _ptr(running_mean), _ptr(running_var), _ptr(x), _ptr(residual),
            _ptr(gamma), _ptr(beta), b, c, h * w, float(eps), float(momentum), int(bool(relu)), _ptr(mask),
            _stream())), "sis_bn_fused_fwd")
    return y, mean, invstd, mask


def bn_act_fwd(x, residual, mean, invstd, gamma, beta, relu, want_mask=False):
    """y = [relu](bn(x) [+ residual]); ``want_mask`` (with relu): also the 1-bit-per-element sign mask the backward can read
    instead of y -> (y, mask)."""
    b, c, h, w = x.shape
    y = torch.empty_like(x)
    mask = torch.empty(lib().sis_bn_mask_words(b, c, h * w), dtype=torch.int64, device=x.device) if (want_mask and relu) else None
    with torch.cuda.device(x.device):
        _check(lib().sis_bn_act_fwd(_ptr(y), _ptr(x), _ptr(residual), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), b, c,
                                    h * w, int(bool(relu)), _ptr(mask), _stream()), "sis_bn_act_fwd")
    return (y, mask) if want_mask else y


def bn_act_bwd(dy, y, x, mean, invstd, gamma, relu, want_residual_grad, mask=None):
    """``mask`` (from ``bn_act_fwd(..., want_mask=True)``) replaces ``y`` as the ReLU gate (y may be None)."""
    b, c, h, w = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_residual_grad else None
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib().sis_bn_workspace_floats(b, c, h * w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        # (kernel name reported by the library: bn_wide_bwd_kernel / bn_fused_bwd_kernel where a channel fits one workgroup, else the
        # three-launch form)
        _check(_launch(None, 0.0, 4.0 * (3 + want_residual_grad) * x.numel(), lambda: lib().sis_bn_act_bwd(
            _ptr(dx), _ptr(dres), _ptr(dgamma), _ptr(dbeta), _ptr(dy), _ptr(y), _ptr(x), _ptr(mean),
            _ptr(invstd), _ptr(gamma), _ptr(ws), b, c, h * w, int(bool(relu)), _ptr(mask), _stream())),
               "sis_bn_act_bwd")
    return dx, dres, dgamma, dbeta
