"""ctypes binding of the C-ABI library `csrc/libnndet_amd.so` (declared in include/nndet_amd.h).

There is NO fallback: if the library is missing or a call fails, an exception is raised. PyTorch is
only used for device memory (caching allocator) and the current stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NNDET_AMD_LIB: another build of the same library (A/B measurements of two kernel versions in one gpurun call, tools/gpu_round.sh)
LIB_PATH = os.environ.get("NNDET_AMD_LIB") or os.path.join(_HERE, "csrc", "libnndet_amd.so")

F32, BF16, F16 = 0, 1, 2
STATS_REPLICAS = 32
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}


class NndetError(RuntimeError):
    pass


class NndetConv(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("transposed", C.c_int32), ("batch", C.c_int32),
                ("cin", C.c_int32), ("cout", C.c_int32), ("cin_p", C.c_int32), ("cout_p", C.c_int32),
                ("in_d", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
                ("out_d", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32),
                ("k", C.c_int32 * 3), ("s", C.c_int32 * 3), ("p", C.c_int32 * 3),
                ("in_affine", C.c_void_p), ("in_relu", C.c_int32), ("reserved_", C.c_int32)]


HEAD_MAX_LEVELS = 8


class NndetHeadLevels(C.Structure):
    _fields_ = [("nlev", C.c_int32), ("reserved_", C.c_int32),
                ("y", C.c_void_p * HEAD_MAX_LEVELS), ("dy", C.c_void_p * HEAD_MAX_LEVELS),
                ("scale", C.c_void_p * HEAD_MAX_LEVELS), ("dscale", C.c_void_p * HEAD_MAX_LEVELS),
                ("points", C.c_int64 * HEAD_MAX_LEVELS)]


MAX_ITEMS = 32


class NndetItems(C.Structure):
    """Ragged batch: items of different spatial size in one [rows, C_p] buffer (include/nndet_amd.h)."""
    _fields_ = [("n_items", C.c_int32), ("reserved_", C.c_int32),
                ("dims", (C.c_int32 * 3) * MAX_ITEMS), ("row_off", C.c_int64 * MAX_ITEMS)]


_P, _I64, _I32, _F, _SZ = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t
_CONVP = C.POINTER(NndetConv)
_ITEMSP = C.POINTER(NndetItems)

# name -> (restype, argtypes); every symbol include/nndet_amd.h declares
SIGNATURES = {
    "nndet_version": (C.c_char_p, []),
    "nndet_arch": (C.c_char_p, []),
    "nndet_nms3d_workspace_bytes": (_SZ, [_I64]),
    "nndet_nms3d_f32": (C.c_int, [_P, _P, _I64, _F, _P, _P, _P, _SZ, _P]),
    "nndet_nms2d_f32": (C.c_int, [_P, _P, _I64, _F, _P, _P, _P, _SZ, _P]),
    "nndet_nms3d_sorted_f32": (C.c_int, [_P, _P, _I64, _F, _P, _P, _P, _SZ, _P]),
    "nndet_iou3d_pairwise_f32": (C.c_int, [_P, _I64, _P, _I64, _F, _P, _P]),
    "nndet_giou3d_pairwise_f32": (C.c_int, [_P, _I64, _P, _I64, _F, _P, _P]),
    "nndet_iou3d_rowmax_f32": (C.c_int, [_P, _I64, _P, _I64, _F, _P, _P]),
    "nndet_stream_create_cumask": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_void_p)]),
    "nndet_stream_destroy": (C.c_int, [_P]),
    "nndet_probe_wgrad3d": (C.c_int, [_I64, _P, _P]),
    "nndet_giou3d_pairwise_bwd_f32": (C.c_int, [_P, _I64, _P, _I64, _P, _F, _P, _P, _P]),
    "nndet_giou3d_diag_fwd_f32": (C.c_int, [_P, _P, _I64, _F, _P, _P]),
    "nndet_giou3d_diag_bwd_f32": (C.c_int, [_P, _P, _P, _I64, _F, _P, _P]),
    "nndet_anchors3d_grid_f32": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "nndet_atss3d_workspace_bytes": (_SZ, [_I64, _I64, _I32, _I32]),
    "nndet_atss3d_match_f32": (C.c_int, [_P, _I64, _P, _I64, C.POINTER(C.c_int64), _I32, _I32, _P, _P, _SZ, _P]),
    "nndet_atss3d_match_batched_f32": (C.c_int, [_P, _I64, C.POINTER(C.c_int32), _I32, _P, _I64, C.POINTER(C.c_int64), _I32, _I32, _P, _P, _SZ, _P]),
    "nndet_iou_match3d_f32": (C.c_int, [_P, _I64, _P, _I64, _F, _F, _I32, _P, _P, _SZ, _P]),
    "nndet_atss3d_assign_batched_f32": (C.c_int, [_P, _P, _I64, C.POINTER(C.c_int32), _I32, _P, _I64, C.POINTER(C.c_int64), _I32, _I32, _I32, _F, _P, _P,
                                                  _P, _SZ, _P]),
    "nndet_decode_clip3d_f32": (C.c_int, [_P, _P, _I64, _I64, _F, _F, _F, _F, _P, _P]),
    "nndet_postprocess3d_workspace_bytes": (_SZ, [_I32, _I64, _I32, _I32]),
    "nndet_postprocess3d_f32": (C.c_int, [_P, _I32, _P, _P, _I32, _I64, _I32, _F, _F, _F, _F, _I32, _F, _I32, _F, _I32, _F, _I32,
                                          _P, _P, _P, _P, _P, _SZ, _P]),
    "nndet_postprocess3d_rows_f32": (C.c_int, [_P, _P, _P, _I32, _I64, _F, _F, _F, _I32, _F, _I32, _F, _I32, _F, _I32,
                                               _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "nndet_instances_to_targets_f32": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nndet_hnm_sample_workspace_bytes": (_SZ, [_I32, C.c_double, _I32, C.c_double]),
    "nndet_hnm_neg_capacity": (_I32, [_I32, C.c_double, _I32]),
    "nndet_head_gather_f32": (C.c_int, [_I32, C.POINTER(NndetHeadLevels), _I32, _I32, _I32, _P, _P]),
    "nndet_head_gather_backward": (C.c_int, [_I32, C.POINTER(NndetHeadLevels), _I32, _I32, _I32, _P, _P]),
    "nndet_hnm_sample_f32": (C.c_int, [_P, _P, _I32, _I64, _I32, _I32, C.c_double, _I32, C.c_double, C.c_uint64, _I32, _P, _P, _P,
                                       _P, _SZ, _P]),
    "nndet_detloss_f32": (C.c_int, [_P, _P, _P, _I32, _P, _I32, _P, _P, _P, _P, _I64, _I32, _F, _F, _F, _I32, _F, _I32, _P, _P, _P, _P]),
    "nndet_detloss_compact_f32": (C.c_int, [_P, _P, _P, _I32, _P, _I32, _P, _P, _P, _P, _I64, _I32, _F, _F, _F, _I32, _F, _I32, _P, _P, _P, _P]),
    "nndet_detloss_matched_f32": (C.c_int, [_P, _P, _I32, _P, _I32, _P, _I32, _P, _P, _P, _P, C.POINTER(C.c_int32), _I32, _P, _I64, _I32, _F, _F, _F,
                                            _I32, _F, _I32, _P, _P, _P, _P]),
    "nndet_detloss_scatter_f32": (C.c_int, [_P, _I32, _P, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    "nndet_detloss_scatter2_f32": (C.c_int, [_P, _I32, _P, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nndet_wbc3d_workspace_bytes": (_SZ, [_I64]),
    "nndet_wbc3d_f32": (C.c_int, [_P, _P, _P, _P, _P, _I64, _F, _F, _I32, _F, _P, _P, _P, _P, _P, _SZ, _P]),
    "nndet_packed_weight_elems": (_SZ, [_CONVP, _I32]),
    "nndet_pack_weight": (C.c_int, [_CONVP, _I32, _P, _P, _P]),
    "nndet_pack_weights_batched": (C.c_int, [C.POINTER(NndetConv), C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I32, _P]),
    "nndet_stem_block_supported": (_I32, [_CONVP]),
    "nndet_stem_block_forward": (C.c_int, [_CONVP, _P, _P, _P, _P, _F, _I32, _P, _P, _P, _P]),
    "nndet_stem_block_backward_workspace_bytes": (_SZ, [_CONVP]),
    "nndet_stem_block_backward": (C.c_int, [_CONVP, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _SZ, _P]),
    "nndet_conv3d_forward": (C.c_int, [_CONVP, _P, _P, _P, _P, _P, _P, _P]),
    "nndet_conv3d_backward_data": (C.c_int, [_CONVP, _P, _P, _P, _P]),
    "nndet_conv3d_splitk_workspace_bytes": (_SZ, [_CONVP, _I32]),
    "nndet_conv3d_forward_ws": (C.c_int, [_CONVP, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "nndet_conv3d_forward_norm_input_fused": (C.c_int32, [_CONVP]),
    "nndet_conv3d_forward_norm_input": (C.c_int, [_CONVP, _P, _P, _P, _P, _P, _P, _P]),
    "nndet_conv3d_backward_data_ws": (C.c_int, [_CONVP, _P, _P, _P, _P, _SZ, _P]),
    "nndet_conv3d_dgrad_fuses_bias": (_I32, [_CONVP]),
    "nndet_conv3d_backward_data_bias": (C.c_int, [_CONVP, _P, _P, _P, _P, _P]),
    "nndet_conv3d_backward_data_acc": (C.c_int, [_CONVP, _P, _P, _P, _P, _P]),
    "nndet_conv3d_dgrad_fuses_norm_reduce": (_I32, [_CONVP]),
    "nndet_conv3d_backward_data_acc_normred": (C.c_int, [_CONVP, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P, _P]),
    "nndet_conv3d_backward_data_normred": (C.c_int, [_CONVP, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P, _P]),
    "nndet_conv3d_dgrad_normred_supported": (C.c_int32, [_CONVP]),
    "nndet_conv3d_wgrad_workspace_bytes": (_SZ, [_CONVP]),
    "nndet_conv3d_backward_weight": (C.c_int, [_CONVP, _P, _P, _P, _P, _P, _SZ, _P]),
    "nndet_conv3d_forward_items": (C.c_int, [_CONVP, _ITEMSP, _P, _P, _P, _P, _P, _P]),
    "nndet_conv3d_backward_data_items": (C.c_int, [_CONVP, _ITEMSP, _P, _P, _P, _P]),
    "nndet_conv3d_backward_weight_items": (C.c_int, [_CONVP, _ITEMSP, _P, _P, _P, _P, _P, _SZ, _P]),
    "nndet_norm_apply_items": (C.c_int, [_I32, _P, _P, _P, _P, _ITEMSP, _I32, _I32, _I32, _F, _I32, _P, _P, _P]),
    "nndet_norm_backward_items": (C.c_int, [_I32, _P, _P, _P, _P, _P, _ITEMSP, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "nndet_norm_apply_items_split": (C.c_int, [_I32, _P, _P, _P, _P, _ITEMSP, _I32, _I32, _I32, _F, _I32, _P, _P, _I32, _P, _P]),
    "nndet_norm_backward_items_split": (C.c_int, [_I32, _P, _P, _P, _I32, _P, _P, _P, _ITEMSP, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "nndet_norm_stats": (C.c_int, [_I32, _P, _I32, _I64, _I32, _P, _P]),
    "nndet_norm_finalize": (C.c_int, [_P, _P, _P, _I32, _I64, _I32, _I32, _I32, _F, _P, _P, _P]),
    "nndet_affine_apply": (C.c_int, [_I32, _P, _P, _I32, _I64, _I32, _I32, _P, _P]),
    "nndet_norm_apply": (C.c_int, [_I32, _P, _P, _P, _P, _I32, _I64, _I32, _I32, _I32, _F, _I32, _P, _P, _P]),
    "nndet_norm_backward": (C.c_int, [_I32, _P, _P, _P, _P, _P, _I32, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "nndet_norm_backward_presummed": (C.c_int, [_I32, _P, _P, _P, _P, _P, _I32, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "nndet_segloss_forward": (C.c_int, [_I32, _P, _P, _I64, _I32, _P, _P]),
    "nndet_segloss_backward": (C.c_int, [_I32, _P, _P, _I64, _I32, _P, _P, _P]),
    "nndet_seghead_forward": (C.c_int, [_I32, _P, _I32, _I32, _P, _P, _P, _I64, _P, _P]),
    "nndet_seghead_backward": (C.c_int, [_I32, _P, _I32, _I32, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "nndet_seghead_backward_rank1": (C.c_int, [_I32, _P, _I32, _I32, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "nndet_segbranch_replicas": (C.c_int, []),
    "nndet_segbranch_forward": (C.c_int, [_I32, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    "nndet_segbranch_forward2": (C.c_int, [_I32, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "nndet_segbranch_forward_up": (C.c_int, [_I32, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "nndet_segbranch_s2d": (C.c_int, [_I32, _P, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "nndet_segbranch_backward": (C.c_int, [_I32, _P, _P, _I64, _P, _P, _P, _P]),
    "nndet_segbranch_compose_up": (C.c_int, [_I32] + [_P] * 10 + [_I32] + [_P] * 9),
    "nndet_segbranch_param_grads": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _P, _P]),
    "nndet_head_out_sparse_scatter": (C.c_int, [_I32, C.POINTER(NndetHeadLevels), _I32, _I32, _I32, C.POINTER(C.c_int64), _P, _P, _I32,
                                               _P, _I32, _P, _P, _P, _P, _P]),
    "nndet_conv_out_sparse_scale_backward": (C.c_int, [C.POINTER(NndetHeadLevels), _P, _P, _P, _I32, _I32, _P, _P]),
    "nndet_conv_out_sparse_forward": (C.c_int, [_CONVP, _ITEMSP, C.POINTER(NndetHeadLevels), _I32, _I32, _I32, C.POINTER(C.c_int64), _P,
                                               _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nndet_conv_out_sparse_backward": (C.c_int, [_CONVP, _ITEMSP, _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "nndet_segloss_tail_f32": (C.c_int, [_P, _I64, _F, _F, _F, _P, _P, _P]),
    "nndet_sigmoid_max_f32": (C.c_int, [_P, _I64, _I32, _P, _P]),
}

_lib = None


def load():
    """Load the shared library (once). Raises NndetError if it is absent: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise NndetError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             f"or nndetection_amd/csrc/build.sh (hipcc --offload-arch=gfx950). No CPU fallback exists.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise NndetError(f"unsupported activation dtype {t.dtype}; use float32, bfloat16 or float16")


def autocast_input(x: torch.Tensor) -> torch.Tensor:
    """B2: under `torch.autocast` the reference's convolutions run in the autocast dtype whatever the dtype of their
    input (nndet/arch/conv.py:54-143 under pl.Trainer(precision=16)). The HIP kernels take their arithmetic type from the
    activation tensor, so an fp32 CUDA activation that enters a conv block / the network inside an autocast region is cast to the
    autocast dtype first (a differentiable torch cast); 16-bit inputs and everything outside autocast pass through unchanged."""
    if x.is_cuda and x.dtype == torch.float32 and torch.is_autocast_enabled():
        get = getattr(torch, "get_autocast_dtype", None)
        dt = get("cuda") if get is not None else torch.get_autocast_gpu_dtype()
        if dt in _DT:
            return x.to(dt)
    return x


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise NndetError("nndetection_amd kernels need tensors on the GPU (no CPU path)")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device (the raw getter costs ~0.3 us; torch.cuda.current_stream()
    builds a Stream object: ~10 us, at ~290 calls per training step)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "workspace too small"}.get(rc, f"hipError {rc}")
        raise NndetError(f"{what} failed: {kind}")


_workspaces = {}


def workspace(nbytes: int, device, raw_stream=None) -> "torch.Tensor":
    """One grow-only scratch buffer per (device, stream): launches on one stream are serialised, so they can share it;
    branches that run concurrently on side streams (the detection head levels) get their own. raw_stream: the stream the
    launch goes to if it is not torch's current one (the weight-gradient stream)."""
    key = (device.type, device.index, (raw_stream if raw_stream is not None else torch.cuda.current_stream(device).cuda_stream)
           if device.type == "cuda" else 0)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and raw_stream is not None and device.type == "cuda":
            # the buffer being replaced was allocated under torch's current stream but is USED on `raw_stream` (the lagging
            # weight-gradient stream): tell the caching allocator, or the next allocation on the current stream could alias it while
            # a weight-gradient kernel still writes its split-K partials (only while the workspace is still growing)
            buf.record_stream(torch.cuda.ExternalStream(raw_stream, device=device))
        buf = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


class _ZeroArena:
    """Zero-initialised scratch for tensors that die inside one forward / backward call (norm statistics and reduction
    buffers: 2 per normalised conv and step). Bump allocation from one buffer that is re-zeroed with ONE fill per step
    (`reset()`, called at the start of BaseRetinaNet.forward) instead of one fill kernel per tensor. Memory that was handed
    out and not yet reset is never handed out again; when the buffer is exhausted (no reset, e.g. a bare conv module in a
    loop) the caller falls back to torch.zeros, so results never depend on reset() being called."""

    def __init__(self, device, nbytes=64 << 20):
        self.buf = torch.zeros((nbytes,), dtype=torch.uint8, device=device)
        self.off = 0

    def take(self, nbytes):
        start = (self.off + 255) // 256 * 256
        if start + nbytes > self.buf.numel():
            return None
        self.off = start + nbytes
        return self.buf[start:start + nbytes]

    def reset(self):
        if self.off:
            self.buf[:self.off].zero_()
            self.off = 0


_arenas = {}


def arena_reset(device):
    a = _arenas.get((device.type, device.index))
    if a is not None:
        a.reset()


_NO_ARENA = bool(os.environ.get("NNDET_NO_ARENA"))      # (read at import: the function below runs ~50 times per training step)


def arena_zeros(shape, dtype, device):
    """Zeroed tensor for a temporary that is dead when the current autograd node returns (see _ZeroArena)."""
    if device.type != "cuda" or _NO_ARENA:
        return torch.zeros(shape, dtype=dtype, device=device)
    key = (device.type, device.index)
    a = _arenas.get(key)
    if a is None:
        a = _arenas[key] = _ZeroArena(device)
    n = 1
    for s in shape:
        n *= int(s)
    raw = a.take(n * torch.empty((), dtype=dtype).element_size())
    if raw is None:
        return torch.zeros(shape, dtype=dtype, device=device)
    return raw.view(dtype).view(shape)


class _GradPool:
    """One zero-filled fp32 buffer per training step for ALL parameter-gradient accumulators of the convolution nodes (54 fills
    of a few KB .. 11 MB per step otherwise). A fresh torch tensor per step: the gradients handed to autograd are views of it and
    keep it alive exactly as long as they live, so nothing depends on when the optimizer drops them.

    STATIC layout (round 5, installed by nndetection_amd.ddp.GradAllReducer): every parameter owns a fixed region of the
    reducer's flat bucket memory, `begin()` zero-fills that memory (one fill, as before) and `take_for` hands a node the regions of
    ITS parameters, so the gradients autograd adopts already sit where the all-reduce reads and writes: no bucket copy, and the
    region of a parameter without a gradient on this rank (the regressor on a rank without positives, decoder.out.P1) is zero by
    construction. A region is handed out once per step and only while the parameter has no `.grad` (a second node of the same
    parameter, or a second backward pass without zero_grad, gets fresh memory and autograd adds it in)."""

    def __init__(self):
        self.buf, self.off = None, 0
        self.static, self.static_flat, self.owner, self.taken, self.armed = None, None, None, set(), False

    def install_static(self, views: dict, flat: torch.Tensor, owner=None) -> None:
        """views: {parameter: 1-D fp32 view of `flat` with the parameter's numel} (keyed by the Parameter OBJECTS: a hit is the
        same live object, never a recycled id); owner: the module whose forward pass calls `begin(..., owner=module)` -- the layout
        is only armed for that module's steps (another model in the same process keeps the per-step pool)."""
        import weakref
        self.static, self.static_flat, self.taken, self.armed = dict(views), flat, set(), False
        self.owner = weakref.ref(owner) if owner is not None else None

    def remove_static(self, flat=None) -> None:
        if flat is None or flat is self.static_flat:
            self.static, self.static_flat, self.owner, self.taken, self.armed = None, None, None, set(), False

    def _static_in_use(self) -> bool:
        """Does a parameter's `.grad` still live in the static memory? After a backward pass (and after the reducer's finish()) the
        gradients ARE that memory: a zero fill would wipe them -- gradient accumulation over micro-batches without zero_grad, a
        grad-enabled forward pass between finish() and optimizer.step()."""
        flat = self.static_flat
        lo = flat.data_ptr()
        hi = lo + flat.numel() * flat.element_size()
        for p in self.static:
            g = p.grad
            if g is not None and lo <= g.data_ptr() < hi:
                return True
        return False

    def begin(self, numel: int, device, owner=None):
        self.armed = False
        # the static layout is armed (and its memory zero-filled) only by a TOP-LEVEL forward pass of the owner whose parameters hold
        # no gradient in that memory: a forward pass inside a backward pass (checkpoint recompute) or on top of accumulated / reduced
        # gradients keeps the per-step pool, and autograd adds its gradients to the existing ones
        if self.static is not None and self.static_flat.device == device and (self.owner is None or self.owner() is owner) \
                and graph_task_id() == -1 and not self._static_in_use():
            for fn in list(step_begin_listeners):
                fn()
            self.static_flat.zero_()                 # (the previous step's all-reduce was waited for by finish() on this stream)
            self.taken, self.armed = set(), True
            self.buf, self.off = None, 0             # the few nodes that cannot use their region take fresh zeroed memory
            return
        self.buf = torch.zeros((int(numel),), dtype=torch.float32, device=device)
        self.off = 0

    def take(self, numel: int, device):
        b = self.buf
        start = (self.off + 63) // 64 * 64
        if b is None or b.device != device or start + numel > b.numel():
            return torch.zeros((numel,), dtype=torch.float32, device=device)
        self.off = start + numel
        return b[start:start + numel]

    def take_for(self, parts, device):
        """parts: [(parameter or None, numel), ...] of ONE autograd node -> a zeroed fp32 1-D tensor per part: the parameter's
        static region where one is installed and usable, else consecutive slices of one `take` (the round-2 behaviour)."""
        out = [None] * len(parts)
        if self.armed and self.static is not None:
            for i, (p, n) in enumerate(parts):
                v = self.static.get(p) if p is not None else None
                if v is not None and n == v.numel() and v.device == device and p.grad is None and id(p) not in self.taken:
                    self.taken.add(id(p))
                    out[i] = v.view(-1)              # a FRESH view: autograd adopts a gradient only if nobody else holds the tensor object
        rest = [i for i in range(len(parts)) if out[i] is None]
        if rest:
            g = self.take(sum(int(parts[i][1]) for i in rest), device)
            o = 0
            for i in rest:
                n = int(parts[i][1])
                out[i] = g[o:o + n]
                o += n
        return out

    def end(self):
        self.buf, self.off = None, 0


grad_pool = _GradPool()

# Callbacks `fn(list_of_parameters)` told, during the forward pass, that these parameters will NOT receive a gradient in the coming
# backward pass (arch/heads.py: the regressor on the compact loss route when the batch has no positive anchor). The gradient reducer
# (nndetection_amd.ddp) registers one: it may then launch their bucket from the hooks of the other parameters instead of from finish().
no_grad_listeners = []
# Callbacks `fn()` run when a top-level training forward pass arms the static gradient layout, i.e. when a new step begins: the reducer
# drops the per-step declarations of a forward pass that was never followed by backward + finish() (skipped step, exception, NaN guard).
step_begin_listeners = []


def notify_no_grad(params) -> None:
    for fn in list(no_grad_listeners):
        fn(params)


_aux_streams = {}


def aux_stream(device):
    """One auxiliary stream per device for work that has to leave the critical chain of the forward pass (arch/conv.py: the
    materialising norm pass of an activation whose first consumer reads the pre-norm tensor)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _aux_streams.get(idx)
    if st is None:
        st = _aux_streams[idx] = new_stream("libaux", device)
    return st


# Two private torch hooks carry the multi-stream backward pass: the id of the running graph task (to tell one backward pass from the next,
# and a forward pass that runs INSIDE one) and the engine's end-of-pass callback queue. Both are probed once; without them the
# weight-gradient stream is switched off (everything stays on the calling stream: slower, same results) instead of failing mid-step
# after a torch upgrade.
_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)
_engine = getattr(getattr(torch.autograd, "Variable", None), "_execution_engine", None)
_queue_callback = getattr(_engine, "queue_callback", None)
PRIVATE_AUTOGRAD_HOOKS = _graph_task_id is not None and _queue_callback is not None
if not PRIVATE_AUTOGRAD_HOOKS:
    import warnings
    warnings.warn("nndetection_amd: torch._C._current_graph_task_id / the autograd engine's queue_callback are not available in this "
                  "torch build; weight gradients stay on the main stream (NNDET_WGRAD_STREAM=0 behaviour)")


def graph_task_id() -> int:
    """Id of the backward pass that is running on this thread, -1 outside of one (or when torch does not tell)."""
    return _graph_task_id() if _graph_task_id is not None else -1


_cumask_streams = []                                 # (keeps the raw handles alive for the life of the process)


def cumask_stream(dev, spec: str):
    """A stream restricted to a CU partition (hipExtStreamCreateWithCUMask through the C ABI), as a torch.cuda.ExternalStream.
    spec: "N" = CUs [0, N) of the mask order; "N:S" = N of every S consecutive CUs (an interleaved partition); "xHEX" = the literal
    256-bit mask. Measured in round 6: a CU-masked queue runs the training step 3 x slower whatever the mask (profiles/round6_ab_cumask.txt)."""
    words = (C.c_uint32 * 8)()
    if spec.startswith("x"):
        v = int(spec[1:], 16)
        for w in range(8):
            words[w] = (v >> (32 * w)) & 0xffffffff
    elif ":" in spec:                                # "N:S" = N of every S consecutive CUs (an interleaved partition)
        n, st = (int(t) for t in spec.split(":"))
        for i in range(256):
            if i % st < n:
                words[i // 32] |= 1 << (i % 32)
    else:
        n = max(1, min(256, int(spec)))
        for i in range(n):
            words[i // 32] |= 1 << (i % 32)
    out = C.c_void_p()
    with torch.cuda.device(dev):
        check(load().nndet_stream_create_cumask(C.cast(words, C.c_void_p), 8, C.byref(out)), "nndet_stream_create_cumask")
    _cumask_streams.append(out.value)
    return torch.cuda.ExternalStream(out.value, device=dev)


# ---- side streams and hardware queues -----------------------------------------------------------------------------------------------
# HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues when they are first used; two streams on one hardware queue are
# serialised in submission order, so WHICH of the step's seven streams share a queue decides how much of their overlap is real: one
# stream created before ours costs 0.65 ms of a 11.65 ms step, four cost 1.6 ms, five or more hardware queues 5.8 ms
# (profiles/round6_stream_mapping.txt). Every side stream of the package is created HERE, in a fixed order and with optional padding
# streams in front of a kind (NNDET_STREAM_PADS="aux:1,tail:0,head:2,wgrad:0,...": the mapping experiments), so that the mapping is the
# package's decision as far as the runtime lets it be one.
_STREAM_PADS = {}
for _kv in os.environ.get("NNDET_STREAM_PADS", "").split(","):
    if ":" in _kv:
        _STREAM_PADS[_kv.split(":")[0].strip()] = int(_kv.split(":")[1])
_pad_streams = []          # (kept alive: a destroyed stream gives its queue reference back)


def new_stream(kind: str, device, priority: int = 0):
    for _ in range(_STREAM_PADS.pop(kind, 0)):               # (once per kind)
        ps = torch.cuda.Stream(device=device, priority=priority)
        with torch.cuda.stream(ps):
            torch.zeros(1, device=device)                    # first use binds the hardware queue
        _pad_streams.append(ps)
    return torch.cuda.Stream(device=device, priority=priority)


class _WgradStreams:
    """Weight-gradient kernels on their own stream. Within a backward pass the data-gradient / norm-backward chain is the critical
    path; a weight gradient is only needed by the optimizer. Launched on a second stream the weight-gradient kernels fill the CUs the
    chain leaves idle -- above all under the deep, small pyramid levels, whose kernels are latency-bound launches of a few
    workgroups (2 ms of a 19 ms step at 5 % of its FLOPs). `side(dev, weight)` returns the stream for one node (or None: stay on the
    current stream); the first use inside a backward pass queues an engine callback that makes the caller's stream wait for the side
    stream when the pass ends, i.e. before the optimizer (or anything else on that stream) reads a gradient. Gradient hooks that read
    gradients DURING the pass (nndetection_amd.ddp) take the stream from `active`. NNDET_WGRAD_STREAM=0 disables it."""

    enabled = os.environ.get("NNDET_WGRAD_STREAM", "1") != "0" and PRIVATE_AUTOGRAD_HOOKS

    def __init__(self):
        self.streams, self.active, self.pending, self.task = {}, {}, [], -1

    def side(self, dev, weight):
        if not self.enabled or dev.type != "cuda":
            return None
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        task = graph_task_id()
        if task != self.task:                        # a new backward pass (the previous one may have died before its callback ran)
            for w in self.pending:
                w._nndet_wg_pending = False
            self.active, self.pending, self.task = {}, [], task
        if weight.grad is not None or getattr(weight, "_nndet_wg_pending", False):
            # the engine will ADD this contribution to an existing gradient on the current stream (a module used twice in one pass,
            # accumulation over several passes): order that add behind the earlier weight-gradient kernels and stay on this stream
            ws = self.active.get(idx)
            if ws is not None:
                torch.cuda.current_stream(dev).wait_stream(ws)
            return None
        ws = self.stream_for(dev)
        if not self.active:
            _queue_callback(self._done)
        self.active[idx] = ws
        weight._nndet_wg_pending = True
        self.pending.append(weight)
        ws.wait_stream(torch.cuda.current_stream(dev))       # dY (and the zero-filled gradient pool) are ready
        return ws

    def stream_for(self, dev):
        """The weight-gradient stream of a device (created on first request)."""
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        ws = self.streams.get(idx)
        if ws is None:
            # lowest queue priority the device offers: the critical chain on the other streams gets the CUs first, the weight
            # gradients take what is left (NNDET_WGRAD_PRIO overrides: -1 high, 0 normal, 1 low on ROCm)
            prio = int(os.environ.get("NNDET_WGRAD_PRIO", "1"))
            mask = os.environ.get("NNDET_WGRAD_CUMASK", "")
            if mask:
                ws = self.streams[idx] = cumask_stream(dev, mask)
            else:
                ws = self.streams[idx] = new_stream("wgrad", dev, prio)
        return ws

    def _done(self):
        for idx, ws in self.active.items():
            torch.cuda.current_stream(idx).wait_stream(ws)
        for w in self.pending:
            w._nndet_wg_pending = False
        self.active, self.pending = {}, []


wgrad_streams = _WgradStreams()


class _GradHints:
    """Side channel between autograd nodes: a node that KNOWS more about the gradient it returns than the dense tensor says (it is
    sparse: arch/heads.py) registers that knowledge under the tensor's address; the node that receives exactly this memory (same
    address and size -- views and no-op casts keep it, anything that copies or sums does not) may use it instead of reading the
    tensor. The dense tensor stays a VALID gradient, so a consumer that finds no hint computes the same result the slow way.
    Why an address key cannot alias: an entry HOLDS its tensor, so the caching allocator cannot hand that address to
    anybody else while the entry exists, and an entry that is gone matches nothing. What the address does not tell is an in-place
    modification -- the engine adds a second contribution in place when it holds the only reference to a gradient: every entry
    therefore carries the tensor's version counter at registration (shared by all views of the storage) and is void when it moved.
    Entries are dropped when consumed, after 16 newer ones, and at the next top-level training forward pass."""

    def __init__(self):
        self.d = {}

    def put(self, t: torch.Tensor, payload: dict) -> None:
        if len(self.d) >= 16:
            self.d.pop(next(iter(self.d)))
        self.d[t.data_ptr()] = (t, payload, t._version)

    def pop(self, t: torch.Tensor):
        e = self.d.pop(t.data_ptr(), None)
        if e is None:
            return None
        keep, payload, ver = e
        if keep.numel() * keep.element_size() != t.numel() * t.element_size() or keep.dtype != t.dtype:
            return None                                  # same address, other tensor (a sub-view): not ours
        if t._version != ver or keep._version != ver:
            return None                                  # written in place since (accumulated by the engine, a hook): the hint is stale
        return payload

    def clear(self) -> None:
        self.d.clear()


grad_hints = _GradHints()


def call(name: str, *args):
    check(getattr(load(), name)(*args), name)
