"""Host wrappers of the library's hand-written GEMMs used by the training backward (no BLAS library on the bf16 path)."""
from __future__ import annotations

import torch

from . import _lib

_workspaces = {}


def _workspace(device, nbytes):
    ws = _workspaces.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[device] = ws
    return ws


def gemm_tn(a: torch.Tensor, b: torch.Tensor, rows: int | None = None, cols: int | None = None, scale: float = 1.0,
            out: torch.Tensor | None = None, accumulate: bool = False) -> torch.Tensor:
    """out[:rows, :cols] (+)= scale * a[:, :rows]^T @ b[:, :cols] in fp32, for bf16 row-major a [E, lda], b [E, ldb] whose row
    strides are multiples of 8 and whose widths, rounded up to 256 (a) / 128 (b), stay inside the row stride
    (egnn_gemm_tn_bf16: the reduction over E runs on the matrix cores, slices of E are added in a fixed order)."""
    if not (a.is_cuda and b.is_cuda) or a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
        raise RuntimeError("gemm_tn needs bf16 CUDA(ROCm) tensors; there is no CPU fallback")
    if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0] or a.stride(1) != 1 or b.stride(1) != 1:
        raise ValueError("gemm_tn: a [E, lda] and b [E, ldb] row-major with the same E")
    E, lda, ldb = a.shape[0], a.stride(0), b.stride(0)
    rows = a.shape[1] if rows is None else rows
    cols = b.shape[1] if cols is None else cols
    M, N = (rows + 255) // 256 * 256, (cols + 127) // 128 * 128
    if M > lda or N > ldb:
        raise ValueError(f"gemm_tn: operand widths padded to {M} / {N} exceed the row strides {lda} / {ldb}")
    L = _lib.lib()
    need = int(L.egnn_gemm_tn_workspace_bytes(E, M, N))
    if need == 0:
        raise ValueError("gemm_tn: unsupported shape")
    ws = _workspace(a.device, need)
    if out is None:
        out = torch.empty(rows, cols, dtype=torch.float32, device=a.device)
        accumulate = False
    if out.dtype != torch.float32 or out.stride(1) != 1 or out.shape[0] < rows or out.shape[1] < cols:
        raise ValueError("gemm_tn: out must be fp32 [>= rows, >= cols] with unit column stride")
    _lib.check(L.egnn_gemm_tn_bf16(_lib.stream_ptr(), E, M, N, _lib.ptr(a), lda, _lib.ptr(b), ldb, float(scale), _lib.ptr(out),
                                   out.stride(0), rows, cols, 1 if accumulate else 0, _lib.ptr(ws), ws.numel()))
    ws.record_stream(torch.cuda.current_stream())
    return out


def pack_rows_weights(w: torch.Tensor, ncols: int | None = None) -> torch.Tensor:
    """fragment pack of an fp32 [K, >= ncols] matrix for ``gemm_rows`` (K % 64 == 0): K * 128 bf16 per chunk of 128 columns"""
    if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 2 or w.stride(1) != 1:
        raise ValueError("pack_rows_weights: fp32 CUDA(ROCm) [K, n] matrix with unit column stride")
    K = w.shape[0]
    ncols = w.shape[1] if ncols is None else ncols
    out = torch.empty((ncols + 127) // 128 * K * 128, dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.lib().egnn_gemm_rows_pack(_lib.stream_ptr(), K, ncols, _lib.ptr(w), w.stride(0), _lib.ptr(out)))
    return out


def gemm_rows(a0: torch.Tensor, w0: torch.Tensor, a1: torch.Tensor | None = None, w1: torch.Tensor | None = None,
              out: torch.Tensor | None = None, k0: int | None = None, chunks: int = 1) -> torch.Tensor:
    """out[e, :128 chunks] = a0[e, :k0] @ W0 (+ a1[e] @ W1), bf16 operands, fp32 accumulate, W as packed by
    ``pack_rows_weights`` (``chunks`` column chunks of 128 in one launch); ``out`` bf16 (default) or fp32, any row stride
    >= 128 chunks (a slice of a wider matrix is fine)  (egnn_gemm_rows_bf16: every operand row is streamed once per chunk)."""
    if not a0.is_cuda or a0.dtype != torch.bfloat16 or a0.dim() != 2 or a0.stride(1) != 1:
        raise RuntimeError("gemm_rows needs bf16 CUDA(ROCm) row-major operands; there is no CPU fallback")
    E = a0.shape[0]
    k0 = a0.shape[1] if k0 is None else k0
    if out is None:
        out = torch.empty(E, 128 * chunks, dtype=torch.bfloat16, device=a0.device)
    if out.dtype not in (torch.bfloat16, torch.float32) or out.shape[0] < E or out.stride(1) != 1 or out.stride(0) < 128 * chunks:
        raise ValueError("gemm_rows: out must be bf16 / fp32 [>= E rows] with a row stride >= 128 per column chunk")
    _lib.check(_lib.lib().egnn_gemm_rows_bf16(_lib.stream_ptr(), E, _lib.ptr(a0), a0.stride(0), k0, _lib.ptr(w0),
                                              _lib.ptr(a1), 0 if a1 is None else a1.stride(0), 0 if a1 is None else a1.shape[1],
                                              _lib.ptr(w1), _lib.ptr(out), out.stride(0), 1 if out.dtype == torch.float32 else 0,
                                              int(chunks)))
    return out


def linear_rows(a: torch.Tensor, weight: torch.Tensor, k: int | None = None) -> torch.Tensor:
    """fp32 [E, n_out] = a[:, :k] @ weight[:, :k]^T for a bf16 row-major ``a`` (k % 64 == 0; columns of ``a`` beyond the weight's
    input width must be zero) and an fp32 nn.Linear weight [n_out, <= k]: the row-streaming kernel, 128 output columns per launch."""
    n_out, n_in = weight.shape
    k = a.shape[1] if k is None else k
    wt = torch.zeros(k, n_out, dtype=torch.float32, device=a.device)
    wt[:n_in] = weight.detach().float().t()
    chunks = (n_out + 127) // 128
    out = torch.empty(a.shape[0], chunks * 128, dtype=torch.float32, device=a.device)
    gemm_rows(a, pack_rows_weights(wt, n_out), out=out, k0=k, chunks=chunks)   # all column chunks in one launch
    return out[:, :n_out]


# ---- fp32-grade products on the bf16 matrix cores: head + remainder operands, three products ---------------
# The backward of the fp32-grade precisions (fp32, bf16x3, f16c8: autograd.py) multiplied through the BLAS library (torch.mm /
# bmm).  These two wrappers run the same products on the library's OWN kernels: every fp32 operand is split into a bf16 head and a
# bf16 remainder (a = a_hi + a_lo, 16 significant bits) and the product is a_lo b_hi + a_hi b_lo + a_hi b_hi with fp32
# accumulation -- what precision bf16x3 does in the forward -- as three calls of gemm_tn (accumulating into one fp32 result) or two
# calls of gemm_rows (its two-operand form carries the small terms).  Relative error 2^-16 per operand (measured against a
# float64 product: tests/test_training.py::test_split_products_match_fp64).
def _split_bf16(t: torch.Tensor, cols: int):
    """fp32 [R, C] -> (head, remainder) bf16 [R, cols] (cols >= C, zero-padded: operand widths of the kernels)"""
    R, C = t.shape
    h = t.to(torch.bfloat16)
    l = (t - h.float()).to(torch.bfloat16)
    if cols == C:
        return h, l
    hi = torch.zeros(R, cols, dtype=torch.bfloat16, device=t.device)
    lo = torch.zeros(R, cols, dtype=torch.bfloat16, device=t.device)
    hi[:, :C] = h
    lo[:, :C] = l
    return hi, lo


def mm_tn_split(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """fp32 [Ma, Nb] = a^T b for fp32 a [E, Ma], b [E, Nb] (the weight-gradient products: reduction over the rows)"""
    Ma, Nb = a.shape[1], b.shape[1]
    Mp, Np = (Ma + 255) // 256 * 256, (Nb + 127) // 128 * 128
    a_hi, a_lo = _split_bf16(a, Mp)
    b_hi, b_lo = _split_bf16(b, Np)
    out = torch.empty(Ma, Nb, dtype=torch.float32, device=a.device)
    gemm_tn(a_lo, b_hi, rows=Ma, cols=Nb, out=out, accumulate=False)   # small terms first
    gemm_tn(a_hi, b_lo, rows=Ma, cols=Nb, out=out, accumulate=True)
    gemm_tn(a_hi, b_hi, rows=Ma, cols=Nb, out=out, accumulate=True)
    return out


def mm_nn_split(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """fp32 [E, N] = a w for fp32 a [E, K], w [K, N] (forward / input-gradient products: every row of a streamed once per 128
    output columns).  ONE launch: the kernel's two-operand form out = A0 W0 + A1 W1 with A0 = [a_hi | a_lo] against [W_hi ; W_hi]
    (both products that share W_hi, reduction width 2 K) and A1 = a_hi against W_lo."""
    E, K = a.shape
    N = w.shape[1]
    Kp = (K + 63) // 64 * 64
    hl = torch.zeros(E, 2 * Kp, dtype=torch.bfloat16, device=a.device) if Kp != K else torch.empty(E, 2 * Kp, dtype=torch.bfloat16, device=a.device)
    hl[:, :K] = a                                    # head (implicit rounding to bf16)
    hl[:, Kp:Kp + K] = a - hl[:, :K].float()         # remainder
    wp = torch.zeros(Kp, N, dtype=torch.float32, device=a.device)
    wp[:K] = w
    w_hi = wp.to(torch.bfloat16).float()
    p0 = pack_rows_weights(torch.cat((w_hi, w_hi), 0).contiguous(), N)
    p1 = pack_rows_weights((wp - w_hi).contiguous(), N)
    chunks = (N + 127) // 128
    direct = out is not None and out.dtype == torch.float32 and out.stride(1) == 1 and out.stride(0) >= 128 * chunks and out.shape[0] >= E
    res = out if direct else torch.empty(E, chunks * 128, dtype=torch.float32, device=a.device)
    gemm_rows(hl, p0, hl[:, :Kp], p1, out=res, k0=2 * Kp, chunks=chunks)
    if direct:
        return out[:, :N]
    if out is not None:
        out.copy_(res[:, :N])
        return out
    return res[:, :N]
