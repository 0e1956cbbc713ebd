"""The host side of libegnn_amd under AddressSanitizer + UBSan, in the build container (no GPU): csrc/host_logic.cpp -- the
schedule builder, the argument / shape validation of the entry points, the padded model dimensions, the split-K plan of the
weight-gradient GEMM, the fork decision -- compiled by `make asan` (g++ -fsanitize=address,undefined) into
build/libegnn_host_asan.so.  The instrumented library needs the ASan runtime loaded first, so every case runs in a child
python with LD_PRELOAD=libasan; a sanitizer report aborts the child (-fno-sanitize-recover, halt_on_error) and fails the test (leak checking is off: the
interpreter itself leaks at exit).
SURVEY section 5 "race detection / sanitizers" stance; GPU sanitizers are not available on this pool."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_LIB = os.path.join(ROOT, "build", "libegnn_host_asan.so")


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _run(body):
    # always through make (a no-op when up to date): a stale instrumented library once hid a test that had fallen behind the header
    subprocess.run(["make", "-C", ROOT, "asan"], check=True, capture_output=True)
    asan = _libasan()
    if asan is None:
        pytest.skip("no libasan in this toolchain")
    prog = textwrap.dedent("""
        import ctypes as C, sys
        import numpy as np
        L = C.CDLL(%r)
        L.egnn_last_error.restype = C.c_char_p
        L.egnn_gemm_tn_workspace_bytes.restype = C.c_size_t
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        OK, EINVAL, ESTATE = 0, -22, -1
    """ % ASAN_LIB) + textwrap.dedent(body) + "\nprint('CHILD-OK')\n"
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CHILD-OK" in r.stdout, f"rc={r.returncode}\nstdout:\n{r.stdout}\nstderr:\n{r.stderr[-4000:]}"
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_schedule_builder_under_sanitizers_matches_goldens():
    _run("""
        G = np.load(%r, allow_pickle=False)
        for tag in ("T1000", "T50", "T200"):
            T, p, s = [float(v) for v in G[tag + ".params"]]
            T = int(T)
            a, sg, tab = (np.zeros(T + 1, np.float32), np.zeros(T + 1, np.float32), np.zeros(4 * (T + 1), np.float32))
            assert L.schedule_table_build(T, C.c_double(s), C.c_double(p), fp(a), fp(sg), fp(tab)) == OK
            assert np.array_equal(a, G[tag + ".alpha"])                      # bit exact vs the executed reference
            assert np.all(np.abs(sg - G[tag + ".sigma"]) <= 1.2e-7 * np.abs(G[tag + ".sigma"]) + 1e-38)
            tab2 = np.zeros_like(tab)
            assert L.schedule_table_from_alpha(T, fp(a), fp(sg), fp(tab2)) == OK and np.array_equal(tab, tab2)
            assert np.all(np.isfinite(tab)) and tab[4 * T + 3] == 1.0
        # every optional output may be absent; T = 1 is the smallest table; odd powers take the powf branch
        one = np.zeros(2, np.float32)
        assert L.schedule_table_build(1, C.c_double(1e-5), C.c_double(2.0), fp(one), None, None) == OK
        assert L.schedule_table_build(7, C.c_double(1e-5), C.c_double(2.5), None, None, None) == OK
        t8 = np.zeros(4 * 8, np.float32)
        assert L.schedule_table_build(7, C.c_double(1e-5), C.c_double(3.0), None, None, fp(t8)) == OK and np.all(np.isfinite(t8))
        # error paths
        assert L.schedule_table_build(0, C.c_double(1e-5), C.c_double(2.0), None, None, None) == EINVAL
        assert b"T must be" in L.egnn_last_error()
        assert L.schedule_table_from_alpha(5, None, None, fp(t8)) == EINVAL
        assert L.schedule_table_from_alpha(0, fp(one), None, fp(t8)) == EINVAL
        a4 = np.array([0.99, 0.9, 0.5, 0.1], np.float32)
        t16 = np.zeros(16, np.float32)
        assert L.schedule_table_from_alpha(3, fp(a4), None, fp(t16)) == OK       # sigma absent: derived from alpha
    """ % os.path.join(ROOT, "tests", "golden", "diffusion_golden.npz"))


def test_model_dims_and_argument_checks_under_sanitizers():
    _run("""
        out = (C.c_int * 10)()
        assert L.egnn_host_model_dims(4, 36, 256, 1024, 1024, 1024, out) == OK
        assert list(out) == [1024, 1024, 256, 1024, 64, 296, 304, 4096, 8, 2]       # WxP WmP MP WhP HP K1P K1Q TC cbx cbm
        assert L.egnn_host_model_dims(2, 3, 5, 7, 300, 130, out) == OK
        assert list(out)[:5] == [512, 256, 256, 256, 32]
        for bad in ((0, 36, 256, 1024, 1024, 1024), (4, 0, 256, 1024, 1024, 1024), (4, 36, 256, 1025, 1024, 1024),
                    (4, 36, 256, 1024, 2048, 1024), (4, 257, 256, 1024, 1024, 1024), (4, 36, 2000, 1024, 1024, 1024)):
            assert L.egnn_host_model_dims(*bad, out) == EINVAL, bad
        assert L.egnn_host_model_dims(4, 36, 256, 1024, 1024, 1024, None) == OK            # the output block is optional
        p = C.c_void_p(64)
        assert L.egnn_host_graph_args_check(1, 8, 16, 2, p, p, p, p, p) == OK
        assert L.egnn_host_graph_args_check(1, 8, 0, 2, None, None, p, p, p) == OK       # E = 0: no edge arrays needed
        assert L.egnn_host_graph_args_check(1, 8, 16, 2, None, p, p, p, p) == EINVAL
        assert L.egnn_host_graph_args_check(1, 0, 16, 2, p, p, p, p, p) == EINVAL
        assert L.egnn_host_graph_args_check(1, 8, -1, 2, p, p, p, p, p) == EINVAL
        assert L.egnn_host_graph_args_check(1, 8, 16, 0, p, p, p, p, p) == EINVAL
        assert L.egnn_host_graph_args_check(0, 8, 16, 2, p, p, p, p, p) == ESTATE
        for prec in (0, 1, 2, 3, 4):                                                     # fp32, bf16, bf16x3, fp16, f16c8 (include/egnn_amd.h)
            assert L.egnn_host_precision_scope_check(1, prec, 0) == OK and L.egnn_host_precision_scope_check(1, prec, 1) == OK
        assert L.egnn_host_precision_scope_check(1, 5, 0) == EINVAL and L.egnn_host_precision_scope_check(1, -1, 0) == EINVAL
        assert L.egnn_host_precision_scope_check(1, 1, 2) == EINVAL and L.egnn_host_precision_scope_check(0, 1, 0) == ESTATE
        assert L.egnn_host_dense_rows_args_check(10, 2048, 32, p, p, p, p) == OK
        assert L.egnn_host_dense_rows_args_check(10, 2049, 32, p, p, p, p) == EINVAL
        assert L.egnn_host_dense_rows_args_check(10, 200, 32, p, None, p, p) == EINVAL
        # fork decision: one 64-atom graph (32 tiles, 64 coordinate workgroups) forks, 16 graphs (full rounds) do not
        assert L.egnn_host_fork_candidate(4032, 1024) == 1 and L.egnn_host_fork_candidate(5 * 4032, 1024) == 1
        assert L.egnn_host_fork_candidate(16 * 4032, 1024) == 0 and L.egnn_host_fork_candidate(0, 1024) == 0
        assert L.egnn_host_fork_candidate(2**31 - 200, 1024) in (0, 1)                   # no signed overflow in the tile count
    """)


def test_gemm_plan_and_4gib_guards_under_sanitizers():
    _run("""
        p = C.c_void_p(64)
        out = (C.c_int * 4)()
        for E, M, N in ((1032192, 1024, 1024), (1032192, 256, 1024), (1032192, 1024, 128), (1, 256, 128), (33, 256, 256),
                        (2**20, 1024, 256), (4095, 512, 384)):
            assert L.egnn_host_plan_gemm_tn(E, M, N, out) == OK
            BN, tn, S, sps = list(out)
            steps = (E + 31) // 32
            assert BN in (128, 256) and tn * BN == N and S >= 1 and sps >= 1
            assert (S - 1) * sps < steps <= S * sps, (E, M, N, list(out))                # every k-step in exactly one slice
            ws = L.egnn_gemm_tn_workspace_bytes(E, M, N)
            assert ws == S * M * N * 4
            assert L.egnn_host_gemm_tn_args_check(E, M, N, p, M, p, N, p, N, M, N, p, C.c_size_t(ws)) == OK
            assert L.egnn_host_gemm_tn_args_check(E, M, N, p, M, p, N, p, N, M, N, p, C.c_size_t(ws - 1)) == EINVAL
        assert L.egnn_gemm_tn_workspace_bytes(100, 255, 128) == 0 and L.egnn_gemm_tn_workspace_bytes(0, 256, 128) == 0
        big = C.c_size_t(1 << 40)
        # ADVICE r03: the last k-step touches rows up to E + 31: E * ld * 2 just under 4 GiB must be refused too
        ld = 1024
        E_edge = (1 << 32) // (ld * 2) - 8                  # E * ld * 2 < 2^32 <= (E + 31) * ld * 2
        assert E_edge * ld * 2 < 2**32 <= (E_edge + 31) * ld * 2
        assert L.egnn_host_gemm_tn_args_check(E_edge, 1024, 1024, p, ld, p, ld, p, 1024, 1024, 1024, p, big) == EINVAL
        assert b"4 GiB" in L.egnn_last_error()
        assert L.egnn_host_gemm_tn_args_check(E_edge - 64, 1024, 1024, p, ld, p, ld, p, 1024, 1024, 1024, p, big) == OK
        assert L.egnn_host_gemm_tn_args_check(1000, 1024, 1024, p, 1000, p, ld, p, 1024, 1024, 1024, p, big) == EINVAL   # lda < M
        assert L.egnn_host_gemm_tn_args_check(1000, 1024, 1024, p, ld, p, ld, None, 1024, 1024, 1024, p, big) == EINVAL
        assert L.egnn_host_gemm_tn_args_check(1000, 1024, 1024, p, ld, p, ld, p, 1024, 1025, 1024, p, big) == EINVAL     # rows > M
        # gemm_rows: 256-row workgroups
        E_edge = (1 << 32) // (ld * 2) - 100
        assert E_edge * ld * 2 < 2**32 <= (E_edge + 255) * ld * 2
        assert L.egnn_host_gemm_rows_args_check(E_edge, p, ld, 1024, p, None, 0, 0, None, p, 128) == EINVAL
        assert L.egnn_host_gemm_rows_args_check(E_edge - 256, p, ld, 1024, p, None, 0, 0, None, p, 128) == OK
        assert L.egnn_host_gemm_rows_args_check(1000, p, ld, 1024, p, p, 256, 256, None, p, 128) == EINVAL            # A1 without W1
        assert L.egnn_host_gemm_rows_args_check(1000, p, ld, 1000, p, None, 0, 0, None, p, 128) == EINVAL             # K % 64
    """)
