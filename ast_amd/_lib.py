"""ctypes binding of libastk.so (include/astk.h).  The product path has no fallback: if the HIP library is
missing or fails to load, importing the compute path raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ASTK_LIB_PATH") or os.path.join(_HERE, "libastk.so")     # (override: timing experiments with debug builds)

TEST_LIB_PATH = os.path.join(_HERE, "libastk_test.so")     # the same library built with -DASTK_TEST_HOOKS (tests only: load_test_hooks)

MAX_CNN = 4
MAX_RNN = 8
MAX_ATTN = 4
# astk.h: arithmetic of an op's products (descriptor field `precision`) and operand mode (`gemm_operands`)
PREC_DEFAULT, PREC_FP16X2, PREC_BF16X3, PREC_F32 = 0, 1, 2, 3
OPERANDS_DEFAULT, OPERANDS_F32, OPERANDS_FP16 = 0, 1, 2
OPERANDS_BY_NAME = {None: OPERANDS_DEFAULT, "default": OPERANDS_DEFAULT, "f32": OPERANDS_F32, "fp16": OPERANDS_FP16}
PREC_BY_NAME = {None: PREC_DEFAULT, "default": PREC_DEFAULT, "fp16x2": PREC_FP16X2, "bf16x3": PREC_BF16X3, "f32": PREC_F32}

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)
c_double_p = C.POINTER(C.c_double)


class _Sized:
    """Descriptors that start with `struct_size` (astk.h): zero-initialised, sized, then the positional / keyword fields BEHIND struct_size."""

    def __init__(self, *a, **k):
        super().__init__()
        names = [f[0] for f in self._fields_[1:]]
        assert len(a) <= len(names)
        for n, v in zip(names, a):
            setattr(self, n, v)
        for n, v in k.items():
            setattr(self, n, v)
        self.struct_size = C.sizeof(type(self))


class CnnDesc(_Sized, C.Structure):
    _fields_ = [("struct_size", C.c_size_t), ("B", C.c_int), ("T", C.c_int), ("D", C.c_int), ("n_layers", C.c_int),
                ("C", C.c_int * MAX_CNN), ("kt", C.c_int * MAX_CNN), ("kf", C.c_int * MAX_CNN),
                ("st", C.c_int * MAX_CNN), ("sf", C.c_int * MAX_CNN), ("pt", C.c_int * MAX_CNN),
                ("bn_eps", C.c_float), ("bn_decay", C.c_float), ("no_bn", C.c_int),
                ("pool_t", C.c_int * MAX_CNN), ("pool_f", C.c_int * MAX_CNN),
                ("precision", C.c_int), ("gemm_operands", C.c_int), ("status_dst", C.c_void_p), ("deterministic", C.c_int)]



class CnnLayerParams(C.Structure):
    _fields_ = [("W", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("avg_mean", C.c_void_p), ("avg_var", C.c_void_p), ("bias", C.c_void_p)]


class CnnLayerGrads(C.Structure):
    _fields_ = [("dW", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dbias", C.c_void_p)]


class LstmStackDesc(_Sized, C.Structure):
    _fields_ = [("struct_size", C.c_size_t), ("T", C.c_int), ("B", C.c_int), ("in_dim", C.c_int), ("h", C.c_int),
                ("n_layers", C.c_int), ("n_dirs", C.c_int), ("out_bound", C.c_float), ("x_amax", C.c_void_p),
                ("precision", C.c_int), ("gemm_operands", C.c_int), ("side_stream", C.c_void_p), ("side_wgs", C.c_int),
                ("deterministic", C.c_int)]



class LstmParams(C.Structure):
    _fields_ = [("Wu", C.c_void_p), ("b", C.c_void_p), ("Wl", C.c_void_p)]


class LstmGrads(C.Structure):
    _fields_ = [("dWu", C.c_void_p), ("db", C.c_void_p), ("dWl", C.c_void_p)]


class RandSeg(C.Structure):
    _fields_ = [("out", C.c_void_p), ("n", C.c_size_t), ("kind", C.c_int), ("a", C.c_float), ("b", C.c_float),
                ("seed", C.c_uint64), ("offset", C.c_uint64)]


RAND_DROPOUT, RAND_NORMAL, RAND_SEG_MAX, RAND_WORDS_MAX = 0, 1, 8, 256


class DecoderDesc(_Sized, C.Structure):
    _fields_ = [("struct_size", C.c_size_t), ("B", C.c_int), ("L", C.c_int), ("T", C.c_int), ("H", C.c_int), ("E", C.c_int),
                ("A", C.c_int), ("V", C.c_int), ("n_layers", C.c_int),
                ("n_attn", C.c_int), ("no_feed_attn", C.c_int), ("ln", C.c_int), ("loss_rows", C.c_int),
                ("use_truth_host", C.POINTER(C.c_int32)), ("precision", C.c_int), ("gemm_operands", C.c_int), ("status_dst", C.c_void_p),
                ("zero_ptr", C.c_void_p), ("zero_bytes", C.c_size_t), ("side_wgs", C.c_int), ("deterministic", C.c_int)]



class DecoderParams(C.Structure):
    _fields_ = [("embed", C.c_void_p), ("lstm", LstmParams * MAX_RNN),
                ("Wa", C.c_void_p), ("ba", C.c_void_p), ("Wc", C.c_void_p), ("bc", C.c_void_p),
                ("Wo", C.c_void_p), ("bo", C.c_void_p), ("class_weight", C.c_void_p),
                ("Wa_x", C.c_void_p * (MAX_ATTN - 1)), ("ba_x", C.c_void_p * (MAX_ATTN - 1)),
                ("ln_gamma", C.c_void_p * MAX_RNN), ("ln_beta", C.c_void_p * MAX_RNN)]


class DecoderGrads(C.Structure):
    _fields_ = [("d_embed", C.c_void_p), ("lstm", LstmGrads * MAX_RNN),
                ("dWa", C.c_void_p), ("dba", C.c_void_p), ("dWc", C.c_void_p), ("dbc", C.c_void_p),
                ("dWo", C.c_void_p), ("dbo", C.c_void_p),
                ("dWa_x", C.c_void_p * (MAX_ATTN - 1)), ("dba_x", C.c_void_p * (MAX_ATTN - 1)),
                ("d_ln_gamma", C.c_void_p * MAX_RNN), ("d_ln_beta", C.c_void_p * MAX_RNN)]


# every symbol include/astk.h declares: name -> (restype, argtypes)
_VP, _I, _L, _SZ, _F, _U64 = C.c_void_p, C.c_int, C.c_long, C.c_size_t, C.c_float, C.c_uint64
SIGNATURES = {
    "astk_version": (C.c_int, []),
    "astk_last_error": (C.c_char_p, []),
    "astk_set_tuning": (C.c_int, [C.c_char_p, C.c_double]),
    "astk_get_tuning": (C.c_int, [C.c_char_p, C.POINTER(C.c_double)]),
    "astk_tuning_key": (C.c_char_p, [_I]),
    "astk_set_low_precision_gemms": (C.c_int, [_I]),
    "astk_get_low_precision_gemms": (C.c_int, []),
    "astk_set_gemm_bf16_split_below": (C.c_double, [C.c_double]),
    "astk_set_gemm_precision": (C.c_int, [_I]),
    "astk_get_gemm_precision": (C.c_int, []),
    "astk_gemm_f32": (C.c_int, [_I, _I, _I, _I, _VP, _L, _VP, _L, _VP, _L, _VP, _I, _I, _I, _L, _L, _L, _VP]),
    "astk_gemm_f32_ex": (C.c_int, [_I, _I, _I, _I, _VP, _L, _VP, _L, _VP, _L, _VP, _I, _I, _I, _L, _L, _L, _I, _VP]),
    "astk_conv_bn_relu_out_dims": (C.c_int, [C.POINTER(CnnDesc), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "astk_conv_bn_relu_workspace_bytes": (_SZ, [C.POINTER(CnnDesc)]),
    "astk_conv_bn_relu_fwd": (C.c_int, [C.POINTER(CnnDesc), C.POINTER(CnnLayerParams), _VP, _VP, _VP, _VP, _SZ, _I, _VP]),
    "astk_conv_bn_relu_bwd": (C.c_int, [C.POINTER(CnnDesc), C.POINTER(CnnLayerParams), C.POINTER(CnnLayerGrads), _VP, _VP, _SZ, _VP]),
    "astk_conv_out_amax": (C.c_void_p, [C.POINTER(CnnDesc), _VP, _SZ]),
    # the exchange callback is passed as an opaque pointer (a ctypes CFUNCTYPE instance converts itself)
    "astk_conv_bn_relu_fwd_sync": (C.c_int, [C.POINTER(CnnDesc), C.POINTER(CnnLayerParams), _VP, _VP, _VP, _VP, _SZ, _I, _VP, _VP, _I, _VP]),
    "astk_conv_bn_relu_bwd_sync": (C.c_int, [C.POINTER(CnnDesc), C.POINTER(CnnLayerParams), C.POINTER(CnnLayerGrads), _VP, _VP, _SZ, _VP, _VP,
                                             _I, _VP]),
    "astk_lstm_stack_workspace_bytes": (_SZ, [C.POINTER(LstmStackDesc)]),
    "astk_lstm_stack_fwd": (C.c_int, [C.POINTER(LstmStackDesc), C.POINTER(LstmParams), _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "astk_lstm_stack_bwd": (C.c_int, [C.POINTER(LstmStackDesc), C.POINTER(LstmParams), C.POINTER(LstmGrads), _VP, _VP, _VP, _VP,
                                      _VP, _VP, _VP, _SZ, _VP]),
    "astk_lstm_stack_bwd_on": (C.c_int, [C.POINTER(LstmStackDesc), C.POINTER(LstmParams), C.POINTER(LstmGrads), _VP, _VP, _VP, _VP,
                                      _VP, _VP, _VP, _SZ, _VP, _VP]),
    "astk_attn_workspace_bytes": (_SZ, [_I, _I, _I]),
    "astk_attn_step_fwd": (C.c_int, [_I, _I, _I, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "astk_attn_step_bwd": (C.c_int, [_I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "astk_decoder_workspace_bytes": (_SZ, [C.POINTER(DecoderDesc)]),
    "astk_decoder_fwd": (C.c_int, [C.POINTER(DecoderDesc), C.POINTER(DecoderParams), _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                   _VP, _SZ, _VP]),
    "astk_decoder_fwd_ex": (C.c_int, [C.POINTER(DecoderDesc), C.POINTER(DecoderParams), _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                      _VP, _SZ, _VP]),
    "astk_decoder_bwd_phase_ex": (C.c_int, [C.POINTER(DecoderDesc), C.POINTER(DecoderParams), C.POINTER(DecoderGrads), _VP, _VP, _VP, _VP,
                                            _VP, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _I, _VP]),
    "astk_layernorm_fwd": (C.c_int, [_I, _I, _VP, _L, _VP, _VP, _F, _VP, _L, _VP]),
    "astk_layernorm_bwd": (C.c_int, [_I, _I, _VP, _L, _VP, _F, _VP, _L, _VP, _L, _VP, _VP, _VP]),
    "astk_step_bn_relu_fwd": (C.c_int, [_I, _I, _I, _VP, _VP, _VP, _VP, _VP, _F, _F, _I, _VP, _VP, _VP]),
    "astk_step_bn_relu_bwd": (C.c_int, [_I, _I, _I, _VP, _VP, _VP, _F, _VP, _VP, _VP, _VP, _VP, _VP]),
    "astk_decoder_bwd": (C.c_int, [C.POINTER(DecoderDesc), C.POINTER(DecoderParams), C.POINTER(DecoderGrads), _VP, _VP, _VP, _VP,
                                   _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "astk_decoder_bwd_phase": (C.c_int, [C.POINTER(DecoderDesc), C.POINTER(DecoderParams), C.POINTER(DecoderGrads), _VP, _VP, _VP, _VP,
                                         _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _I, _VP]),
    "astk_decoder_step_infer": (C.c_int, [C.POINTER(DecoderDesc), C.POINTER(DecoderParams), _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                          _VP, _SZ, _VP]),
    "astk_spin": (C.c_int, [C.c_uint, _VP, _VP]),
    "astk_softmax_ce_fwd": (C.c_int, [_I, _I, _L, _VP, _VP, _L, _VP, _F, _VP, _VP, _VP]),
    "astk_grad_sqnorm": (C.c_int, [_VP, _VP, _F, _SZ, _VP, _VP]),
    "astk_decay_clip_amsgrad_step": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _SZ, _F, _F, _VP, _F, _F, _F, _F, _I, _VP]),
    "astk_decay_clip_sgd_step": (C.c_int, [_VP, _VP, _SZ, _F, _F, _VP, _F, _VP]),
    "astk_grad_sqnorm_scaled": (C.c_int, [_VP, _VP, _F, _F, _SZ, _VP, _VP]),
    "astk_decay_clip_amsgrad_step_scaled": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _SZ, _F, _F, _F, _VP, _F, _F, _F, _F, _I, _VP]),
    "astk_decay_clip_sgd_step_scaled": (C.c_int, [_VP, _VP, _SZ, _F, _F, _F, _VP, _F, _VP]),
    "astk_decay_clip_noise": (C.c_int, [_VP, _VP, _SZ, _F, _F, _F, _VP, _F, _U64, _U64, _VP]),
    "astk_fill_dropout_mask": (C.c_int, [_VP, _SZ, _F, _U64, _U64, _VP]),
    "astk_fill_normal": (C.c_int, [_VP, _SZ, _F, _F, _U64, _U64, _VP]),
    "astk_fill_random": (C.c_int, [C.POINTER(RandSeg), _I, _VP]),
    "astk_fill_random_ex": (C.c_int, [C.POINTER(RandSeg), _I, C.POINTER(C.c_int32), _I, _VP, _VP]),
    "astk_scale_f32": (C.c_int, [_VP, _SZ, _F, _VP]),
    "astk_add_f32": (C.c_int, [_VP, _VP, _SZ, _VP]),
    "astk_bridge_states": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP]),
    "astk_colsum_add_f32": (C.c_int, [_VP, _VP, _L, _I, _I, _VP]),
    "astk_zero_frames": (C.c_int, [_VP, _I, _I, _I, _VP, C.c_double, _U64, _U64, _VP]),
    "astk_zero_frames_draws": (C.c_int, [_I, _I, _VP, C.c_double, _U64, _U64, _VP, _I, _VP, _VP]),
    "astk_persist_status_snapshot": (C.c_int, [_VP, _VP]),
    "astk_persist_status_merge": (C.c_int, [_VP, _VP]),
    "astk_persist_status": (C.c_int, [C.POINTER(C.c_uint), _I]),
    "astk_device_cu_count": (C.c_int, []),
    "astk_lstm_stack_path": (C.c_int, [C.POINTER(LstmStackDesc)]),
    "astk_lstm_stack_free_cus": (C.c_int, [C.POINTER(LstmStackDesc)]),
    "astk_decoder_path": (C.c_int, [C.POINTER(DecoderDesc)]),
    "astk_prof_begin": (C.c_int, []),
    "astk_prof_end": (C.c_int, [C.POINTER(C.c_double)]),
}

# exported by libastk_test.so only (the #ifdef ASTK_TEST_HOOKS sections of include/astk.h)
TEST_HOOK_SIGNATURES = {
    "astk_debug_set_amax_generation": (C.c_int, [C.c_uint]),
    "astk_conv_debug_preact": (C.c_int, [C.POINTER(CnnDesc), _VP, _SZ, _I, _VP, _VP]),
    "astk_conv_debug_kill_units": (C.c_int, [_VP, _I]),
    "astk_debug_gemm_group": (C.c_int, [_I, _I, c_int_p, c_int_p, c_int_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I, _I, _VP]),
}

_lib = None
_test_lib = None


class AstkError(RuntimeError):
    pass


def load():
    """Load libastk.so and bind every declared symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # In a process that also uses PyTorch-ROCm the library must bind to the HIP runtime torch ships (its wheel bundles libamdhip64):
        # loaded first, libastk.so would pull in /opt/rocm's copy beside it, and kernels launched through one runtime on streams of the
        # other fail with "no ROCm-capable device" or crash.  (A pure C / ctypes user without torch is unaffected.)
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise AstkError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU or PyTorch fallback for the compute path)")
    lib = C.CDLL(LIB_PATH)
    _bind(lib, SIGNATURES)
    _lib = lib
    return lib


def _bind(lib, sigs):
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args


class load_test_hooks:
    """Context manager for the tests that need the instrumented build: inside it load() returns libastk_test.so (the same sources
    compiled with -DASTK_TEST_HOOKS, which adds the astk_conv_debug_* / astk_debug_* entry points), so a model built and run inside
    the block executes on the instrumented library.  The product library exports none of those symbols."""

    def __enter__(self):
        global _lib, _test_lib
        if _test_lib is None:
            if not os.path.exists(TEST_LIB_PATH):
                raise AstkError(f"{TEST_LIB_PATH} not found: ast_amd/csrc/build.sh builds it next to libastk.so")
            t = C.CDLL(TEST_LIB_PATH)
            _bind(t, SIGNATURES)
            _bind(t, TEST_HOOK_SIGNATURES)
            _test_lib = t
        self.prev, _lib = _lib, _test_lib
        return _test_lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def set_tuning(key, value):
    """astk_set_tuning (include/astk.h): the library's documented knobs; returns the previous value."""
    lib = load()
    prev = C.c_double()
    check(lib.astk_get_tuning(key.encode(), C.byref(prev)))
    check(lib.astk_set_tuning(key.encode(), float(value)))
    return prev.value


class tuning:
    """`with tuning({"dec.persist": 0}): ...` -- sets knobs of the CURRENTLY loaded library for the block and restores them."""

    def __init__(self, knobs):
        self.knobs = dict(knobs)

    def __enter__(self):
        self.prev = {k: set_tuning(k, v) for k, v in self.knobs.items()}
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_tuning(k, v)
        return False


def check(rc):
    if rc != 0:
        raise AstkError(f"libastk error {rc}: {load().astk_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "libastk expects contiguous tensors"
    return C.c_void_p(t.data_ptr())
