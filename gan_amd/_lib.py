"""ctypes binding of libgan_hip.so (the C ABI declared in include/gan_hip.h).

There is no fallback: if the shared library is missing the import of this module
raises, and every op built on it is unusable.  PyTorch appears here only as the
owner of device memory and of the HIP stream the kernels are queued on.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GAN_HIP_LIB names another build of the same library (the probe build with in-kernel clock stamps, tools/phase_probe.py)
LIB_PATH = os.environ.get("GAN_HIP_LIB") or os.path.join(_HERE, "lib", "libgan_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libgan_hip.so not found at %s: build it with `make -C gan_amd/csrc` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback" % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

c_float_p = ctypes.c_void_p   # raw device pointers travel as integers
c_stream = ctypes.c_void_p


class ConvGeom(ctypes.Structure):
    """gan_conv_geom of include/gan_hip.h."""
    _fields_ = [(k, ctypes.c_int) for k in
                ("n", "hb", "wb", "cb", "hs", "ws", "cs", "kh", "kw", "sh", "sw", "pt", "pl")]

    def key(self):
        return tuple(getattr(self, k) for k, _ in self._fields_)


class BnMovingJob(ctypes.Structure):
    """gan_bn_moving_job of include/gan_hip.h."""
    _fields_ = [("moving_mean", ctypes.c_void_p), ("moving_var", ctypes.c_void_p), ("save_mean", ctypes.c_void_p),
                ("save_rstd", ctypes.c_void_p), ("rows", ctypes.c_int), ("c", ctypes.c_int)]


class BnSaved(ctypes.Structure):
    """gan_bn_saved of include/gan_hip.h."""
    _fields_ = [("x", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("save_mean", ctypes.c_void_p),
                ("save_rstd", ctypes.c_void_p), ("groups", ctypes.c_int), ("act", ctypes.c_int), ("act_param", ctypes.c_float)]


GeomP = ctypes.POINTER(ConvGeom)
P, I, L, F, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); kept in one table so tests can check every symbol of the header
SIGNATURES = {
    "gan_last_error": (ctypes.c_char_p, []),
    "gan_abi_version": (I, []),
    "gan_conv_workspace_bytes": (Z, [GeomP]),
    "gan_conv2d_fwd": (I, [GeomP, P, P, P, P, P, Z, P]),
    "gan_conv2d_fwd_static_ws": (I, [GeomP, P, P, P, P, P]),
    "gan_conv2d_dgrad": (I, [GeomP, P, P, P, P, Z, P]),
    "gan_conv2d_wgrad": (I, [GeomP, P, P, P, P, I, P, Z, P]),
    "gan_deconv2d_fwd": (I, [GeomP, P, P, P, P, P, Z, P]),
    "gan_deconv2d_dgrad": (I, [GeomP, P, P, P, P, Z, P]),
    "gan_deconv2d_wgrad": (I, [GeomP, P, P, P, P, I, P, Z, P]),
    "gan_conv2d_fwd_reflect": (I, [GeomP, P, P, P, P, P, Z, P]),
    "gan_conv2d_wgrad_reflect": (I, [GeomP, P, P, P, P, I, P, Z, P]),
    "gan_conv2d_dgrad_reflect_workspace_bytes": (Z, [GeomP]),
    "gan_conv2d_dgrad_reflect": (I, [GeomP, P, P, P, P, Z, P]),
    "gan_conv2d_fwd_act": (I, [GeomP, P, P, P, P, I, F, F, F, P, Z, P]),
    "gan_deconv2d_fwd_act": (I, [GeomP, P, P, P, P, I, F, F, F, P, Z, P]),
    "gan_conv2d_dgrad_gated": (I, [GeomP, P, P, P, P, I, F, F, F, P, Z, P]),
    "gan_deconv2d_dgrad_gated": (I, [GeomP, P, P, P, P, I, F, F, F, P, Z, P]),
    "gan_linear_workspace_bytes": (Z, [I, I, I]),
    "gan_linear_fwd": (I, [I, I, I, P, P, P, P, P, Z, P]),
    "gan_linear_dgrad": (I, [I, I, I, P, P, P, P, Z, P]),
    "gan_linear_wgrad": (I, [I, I, I, P, P, P, P, I, P, Z, P]),
    "gan_bn_workspace_bytes": (Z, [I, I, I]),
    "gan_bn_fwd_train": (I, [P, P, P, P, P, P, P, P, P, I, I, I, F, F, I, I, F, P, Z, P]),
    "gan_bn_moving_update": (I, [P, P, P, P, I, I, F, F, I, P]),
    "gan_bn_moving_update_multi": (I, [ctypes.POINTER(BnMovingJob), I, F, F, I, P]),
    "gan_bn_fwd_eval": (I, [P, P, P, P, P, P, I, I, F, I, F, P]),
    "gan_bn_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P, Z, P]),
    "gan_bn_tile_blocks": (I, [I, I, I, I]),
    "gan_conv_fwd_bn_stats": (I, [I, GeomP, P, P, P, P, I, P, P, Z, P]),
    "gan_bn_fwd_train_from_stats": (I, [P, P, P, P, P, P, P, P, P, I, I, I, F, F, I, I, F, P, P]),
    "gan_conv_dgrad_bn_sums": (I, [I, GeomP, P, P, P, ctypes.POINTER(BnSaved), P, P, Z, P]),
    "gan_add_bn_sums": (I, [P, P, P, ctypes.POINTER(BnSaved), I, I, P, P]),
    "gan_bn_bwd_tiled": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P, I, P, P]),
    "gan_conv_wgrad_colsum": (I, [I, GeomP, P, P, P, P, I, P, I, P, Z, P]),
    "gan_taps_pack_filter": (I, [P, P, I, I, I, I, I, P]),
    "gan_taps_unpack_filter_grad": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "gan_taps_sum_fwd": (I, [P, P, P, I, I, I, I, I, I, I, I, P]),
    "gan_taps_sum_bwd": (I, [P, P, I, I, I, I, I, I, I, I, P]),
    "gan_taps_spread_bwd": (I, [P, P, I, I, I, I, I, I, I, I, P]),
    "gan_act_fwd": (I, [P, P, L, I, F, F, F, P]),
    "gan_act_bwd": (I, [P, P, P, L, I, F, F, F, P]),
    "gan_add_fwd": (I, [P, P, P, L, P]),
    "gan_concat_fwd": (I, [P, P, P, L, I, I, I, P]),
    "gan_concat_bwd": (I, [P, P, P, L, I, I, I, P]),
    "gan_reflect_pad_fwd": (I, [P, P, I, I, I, I, I, P]),
    "gan_reflect_pad_bwd": (I, [P, P, I, I, I, I, I, P]),
    "gan_sigmoid_bce_mean": (I, [P, L, F, F, I, P, P, P]),
    "gan_mse_const_mean": (I, [P, L, F, F, I, P, P, P]),
    "gan_l1_mean": (I, [P, P, L, F, I, P, P, P]),
    "gan_l1_mean_workspace_bytes": (Z, [L]),
    "gan_l1_mean_ws": (I, [P, P, L, F, I, P, P, P, Z, P]),
    "gan_adam_tf_step": (I, [P, P, P, P, L, F, F, F, F, F, P, P, P]),
    "gan_fill": (I, [P, L, F, P]),
    "gan_conv_plan_set": (I, [I, GeomP, I, I]),
    "gan_conv_plan_get": (I, [I, GeomP, ctypes.POINTER(I), ctypes.POINTER(I), ctypes.POINTER(I)]),
    "gan_conv_plan_clear": (None, []),
    "gan_linear_as_conv_geom": (I, [I, I, I, GeomP]),
    "gan_knob_set": (I, [ctypes.c_char_p, I]),
    "gan_set_exact_f32": (None, [I]),
    "gan_get_exact_f32": (I, []),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)      # AttributeError here = header and library disagree
    _fn.restype = _res
    _fn.argtypes = _args

OP_CONV2D, OP_DECONV2D, OP_CONV2D_REFLECT = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
ACT_CODES = {None: ACT_NONE, 'none': ACT_NONE, 'relu': ACT_RELU, 'lrelu': ACT_LRELU,
             'tanh': ACT_TANH, 'sigmoid': ACT_SIGMOID}


def knob_set(name, value):
    """Set a test / tuning knob of the library (include/gan_hip.h, gan_knob_set)."""
    if lib.gan_knob_set(name.encode(), int(value)) != 0:
        raise ValueError(lib.gan_last_error().decode())


class GanHipError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise GanHipError(lib.gan_last_error().decode())


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Only dense fp32 device tensors are accepted."""
    if t is None:
        return None
    if t.dtype != torch.float32 and t.dtype != torch.int32:
        raise TypeError("libgan_hip takes float32 tensors, got %s" % t.dtype)
    if not t.is_cuda:
        raise GanHipError("libgan_hip needs tensors in GPU memory (got a %s tensor); there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise ValueError("libgan_hip takes dense (contiguous) tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------------------- workspace
_WS = {}
_WS_RETIRED = []    # outgrown buffers stay allocated: captured graphs may still hold their address
_WS_MIN = 64 << 20


def workspace(nbytes):
    """Per-device scratch buffer handed to the library; grown geometrically outside graph capture."""
    dev = torch.cuda.current_device()
    buf = _WS.get(dev)
    if buf is None or buf.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise GanHipError("workspace of %d bytes needed during graph capture; run one eager step first" % nbytes)
        size = max(_WS_MIN, int(nbytes * 1.25))
        if buf is not None:
            _WS_RETIRED.append(buf)
        buf = torch.empty(size, dtype=torch.uint8, device='cuda')
        _WS[dev] = buf
    return buf.data_ptr(), buf.numel()


def conv_geom(n, hb, wb, cb, hs, ws, cs, kh, kw, sh, sw, pt, pl):
    return ConvGeom(n, hb, wb, cb, hs, ws, cs, kh, kw, sh, sw, pt, pl)
