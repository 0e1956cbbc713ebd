"""ctypes binding of libegnn_amd.so (C ABI: include/egnn_amd.h).  No fallback: if the library
is missing or a call fails, an exception is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EGNN_LIB", os.path.join(_HERE, "libegnn_amd.so"))   # EGNN_LIB: timing-experiment builds

PREC_F32, PREC_BF16, PREC_BF16X3, PREC_F16, PREC_F16C8 = 0, 1, 2, 3, 4
NORM_CALL, NORM_GRAPH = 0, 1
PRECISIONS = {"fp32": PREC_F32, "f32": PREC_F32, "float32": PREC_F32, "bf16": PREC_BF16, "bfloat16": PREC_BF16,
              "bf16x3": PREC_BF16X3, "f16c8": PREC_F16C8, "fp16": PREC_F16, "f16": PREC_F16, "float16": PREC_F16}
NORM_SCOPES = {"call": NORM_CALL, "graph": NORM_GRAPH}



_FLAGS_FALLBACK = b"FLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden"


def _flags_line() -> bytes:
    """the compiler flags of the library build: the Makefile's FLAGS line where the Makefile travels with the package, else the
    flags this package's Makefile is known to use (a fingerprint must not fail, or change meaning, because a build file is absent)"""
    try:
        with open(os.path.join(os.path.dirname(_HERE), "Makefile"), "rb") as f:
            flags = [l for l in f.read().splitlines() if l.startswith(b"FLAGS")]
        return b"\n".join(flags)
    except OSError:
        return _FLAGS_FALLBACK


def _sources_sha256(names) -> str:
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    for name in names:
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    h.update(_flags_line())
    return h.hexdigest()


def edge_kernel_sources_sha256() -> str:
    """Fingerprint of the sources the dominant (edge) kernels are compiled from (every header they include, diag.h among them),
    plus the compiler flags: what a PMC measurement of those kernels (profiles/traffic.json) is valid for.  A hash of the .so
    itself would not survive a rebuild in another directory (hipcc derives its per-TU symbol ids from the path)."""
    return _sources_sha256(("edge_x_m16.hip", "edge_bf16_v4.hip", "edge_tile.h", "kernels.h", "common.h", "diag.h"))


def forward_sources_sha256() -> str:
    """Fingerprint of every source an inference forward of any precision runs through (edge kernels of all precisions, node
    kernels, packing): what a measured error table (tools/prec_errors.py) is valid for."""
    return _sources_sha256(("egnn_forward.hip", "edge_x_m16.hip", "edge_bf16_v4.hip", "edge_bf16_v3.hip", "edge_small.hip",
                            "edge_bf16x3.hip", "edge_f16c8.hip", "edge_f16c8w.hip", "edge_f16c8_mphase2.inc", "edge_f16c8_mphase4.inc",
                            "edge_f16c8w_mphase1.inc", "edge_f16c8w_mphase2.inc", "edge_f16c8w_mphasek.inc", "node_bf16.hip", "edge_tile.h",
                            "kernels.h", "common.h", "diag.h"))


def training_sources_sha256() -> str:
    """Fingerprint of everything the HBM traffic of ONE training step depends on: every kernel source and header of the
    library, the compiler flags and the launch sequence (autograd.py, gemm.py): what profiles/traffic_train.json is valid for."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    names = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.cpp")) +
                   glob.glob(os.path.join(csrc, "*.inc")))
    names += [os.path.join(_HERE, "autograd.py"), os.path.join(_HERE, "gemm.py")]
    for path in names:
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    h.update(_flags_line())
    return h.hexdigest()


_vp, _i, _f, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); kept in one table so tests can check every header symbol is exported
SIGNATURES = {
    "egnn_last_error": (C.c_char_p, []),
    "egnn_version": (_i, []),
    "egnn_create": (_i, [C.POINTER(_vp), _i]),
    "egnn_destroy": (_i, [_vp]),
    "egnn_set_model": (_i, [_vp] + [_i] * 6),
    "egnn_pack_layer": (_i, [_vp, _vp, _i] + [_vp] * 16),
    "egnn_set_graph": (_i, [_vp, _i, _i, _i] + [_vp] * 5),
    "egnn_set_side_stream": (_i, [_vp, _vp]),
    "egcl_forward": (_i, [_vp, _vp, _i, _i, _i] + [_vp] * 4),
    "egcl_forward_begin": (_i, [_vp, _vp, _i, _i, _i] + [_vp] * 3),
    "egcl_forward_end": (_i, [_vp, _vp, _i, _i, _i] + [_vp] * 5),
    "egnn_forward": (_i, [_vp, _vp, _i, _i] + [_vp] * 4),
    "egcl_read_aggregates": (_i, [_vp, _vp, _i] + [_vp] * 3),
    "egcl_backward_l1_act": (_i, [_vp, _i, _i, _i] + [_vp] * 7),
    "egcl_backward_heads": (_i, [_vp, _i, _i, _i, _i] + [_vp] * 20),
    "egcl_backward_l1_grad": (_i, [_vp, _i, _i, _i] + [_vp] * 7),
    "egcl_backward_gather_in": (_i, [_vp, _i, _i, _i, _i] + [_vp] * 6),
    "egcl_backward_scatter": (_i, [_vp, _i, _i, _i, _i] + [_vp] * 9),
    "egcl_backward_first_reduce": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i] + [_vp] * 9),
    "egcl_backward_node_act": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "egcl_backward_scatter_geom": (_i, [_vp, _i, _i] + [_vp] * 8),
    "egcl_backward_fused_supported": (_i, [_vp]),
    "egcl_backward_table": (_i, [_vp, _vp, _i, _vp]),
    "egcl_backward_edge_recompute": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i] + [_vp] * 11),
    "egcl_backward_dgrad": (_i, [_vp, _vp, _i, _vp, _i, _i] + [_vp] * 4),
    "egcl_backward_dgrad_reduce": (_i, [_vp, _vp, _i, _vp, _i, _i] + [_vp] * 6),
    "egcl_forward_save": (_i, [_vp, _vp, _i, _i] + [_vp] * 9),
    "egcl_backward_heads_saved": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i] + [_vp] * 10),
    "egnn_gemm_tn_workspace_bytes": (C.c_size_t, [_i, _i, _i]),
    "egnn_gemm_tn_bf16": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _f, _vp, _i, _i, _i, _i, _vp, C.c_size_t]),
    "egnn_gemm_rows_pack": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "egnn_gemm_rows_bf16": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i]),
    "egnn_gamma_tilde": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "egnn_dense_rows": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "egnn_eps": (_i, [_vp, _i, _i, _i, _vp, _i] + [_vp] * 5),
    "egnn_remove_mean": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    "schedule_table_build": (_i, [_i, C.c_double, C.c_double, _fp, _fp, _fp]),
    "schedule_table_from_alpha": (_i, [_i, _fp, _fp, _fp]),
    "ddpm_reverse_step": (_i, [_vp, _i, _i, _i, _vp, _i, _f, _f, _f, _vp, _i, _vp, _vp, _vp, _i]),
    "egnn_sampler_prepare": (_i, [_vp, _i, _i, _f, _vp, _vp, _u64]),
    "egnn_sampler_set_mode": (_i, [_vp, _i]),
    "egnn_sampler_init": (_i, [_vp, _vp, _vp, _vp]),
    "egnn_sampler_run": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "egnn_sampler_final": (_i, [_vp, _vp, _i, _i] + [_vp] * 5),
    "egnn_sampler_state": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_i)]),
    "ddpm_sampler_init": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _u64] + [_vp] * 6),
    "ddpm_sampler_step": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _u64] + [_vp] * 7),
    "ddpm_sampler_final": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _u64] + [_vp] * 10),
    "egnn_fc_graph_build": (_i, [_vp, _i, _i] + [_vp] * 6),
    "egnn_radius_graph_count": (_i, [_vp, _i, _vp, _vp, _vp, _f, _vp]),
    "egnn_radius_graph_fill": (_i, [_vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "egnn_rdf": (_i, [_vp, _i, _vp, _vp, C.c_double, C.c_double, _f, _i, _i, _vp]),
    "egnn_si_o_si": (_i, [_vp, _i, _i, _vp, _vp, _vp, _f, _vp]),
    "egnn_debug_stamps": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "egnn_profile_enable": (_i, [_vp, _i]),
    "egnn_profile_read": (_i, [_vp, _fp, C.POINTER(_i), _fp]),
}

_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build the HIP extension first (`make` at the repo root or "
                "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


class EgnnError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = lib().egnn_last_error()
        raise EgnnError(f"libegnn_amd error {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    import torch
    s = torch.cuda.current_stream() if stream is None else stream
    return C.c_void_p(s.cuda_stream)
