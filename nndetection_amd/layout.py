"""HBM layout of activations: NDHWC (channels-last-3d), channel count padded to a multiple of 32.

Module boundaries exchange LOGICAL [N, C, D, H, W] tensors (the reference's interface); physically they are
views of a contiguous [N, D, H, W, C_p] buffer (C_p = 32 * ceil(C / 32), pad channels are zero). 32 bf16
channels = one 64-byte LDS chunk of the implicit-GEMM kernels, so every voxel row is 16-byte aligned and a
whole number of MFMA K-chunks.
"""
import torch


def cpad(c: int) -> int:
    return (int(c) + 31) // 32 * 32


def mark_padded(t: torch.Tensor) -> torch.Tensor:
    """Flag a logical view whose storage is one of our zero-padded NDHWC buffers."""
    t._nndet_pad_ok = True
    return t


def logical(p: torch.Tensor, c: int) -> torch.Tensor:
    """[N, D, H, W, C_p] contiguous -> logical [N, C, D, H, W] view (no copy)."""
    v = p.permute(0, 4, 1, 2, 3)
    if c != p.shape[4]:
        v = v[:, :c]
    return mark_padded(v)


def _is_ndhwc_view(t: torch.Tensor, cp: int) -> bool:
    N, C, D, H, W = t.shape
    s = t.stride()
    ok = (C == 1 or s[1] == 1) and s[4] == cp and s[3] == W * cp and s[2] == H * W * cp and s[0] == D * H * W * cp
    return ok and (t.storage_offset() * t.element_size()) % 16 == 0


def phys(t: torch.Tensor, dtype: torch.dtype = None, cp: int = None):
    """Logical [N, C, D, H, W] tensor -> (physical [N, D, H, W, C_p] contiguous tensor, C).

    Zero-copy when `t` already is such a view (and either C == C_p or it was produced by our kernels);
    otherwise one conversion pass (permute + zero-pad + cast). C == 1 (the image) is never padded."""
    if t.dim() != 5:
        raise ValueError(f"expected a 5-D [N, C, D, H, W] tensor, got {tuple(t.shape)}")
    N, C, D, H, W = t.shape
    dtype = t.dtype if dtype is None else dtype
    if C == 1 and cp is None:
        return t.to(dtype).contiguous().reshape(N, D, H, W, 1), 1
    cp = cpad(C) if cp is None else cp
    if t.dtype == dtype and _is_ndhwc_view(t, cp) and (C == cp or getattr(t, "_nndet_pad_ok", False)):
        return t.as_strided((N, D, H, W, cp), (D * H * W * cp, H * W * cp, W * cp, cp, 1)), C
    p = torch.zeros((N, D, H, W, cp), dtype=dtype, device=t.device) if C != cp else \
        torch.empty((N, D, H, W, cp), dtype=dtype, device=t.device)
    p[..., :C].copy_(t.permute(0, 2, 3, 4, 1))
    return p, C
