// Host-only logic of libegnn_amd (see host_logic.h).  Plain C++: no HIP call, no device code.
#include "host_logic.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

namespace egnn {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int pow2_ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

int model_dims(int L, int H, int M, int Wm, int Wx, int Wh, ModelDims* out) {
  if (!out || L < 1 || H < 1 || M < 1 || Wm < 1 || Wx < 1 || Wh < 1) { set_error("bad model dims"); return EGNN_EINVAL; }
  if (Wx > 128 * kMaxCB || Wm > 128 * kMaxCB || M > 128 * kMaxCB || H > 32 * kPostMaxOB) {
    set_error("unsupported width: hidden/message widths must be <= %d and H <= %d", 128 * kMaxCB, 32 * kPostMaxOB);
    return EGNN_EINVAL;
  }
  // widths are padded to 256 * 2^k: 8 waves x 32-column blocks in the bf16 kernel, 4 waves x 64 in the fp32 one
  out->WxP = 256 * pow2_ceil((Wx + 255) / 256);   // K and N of mlp_x.2
  out->WmP = 256 * pow2_ceil((Wm + 255) / 256);   // K of mlp_m.2
  out->MP = 256 * pow2_ceil((M + 255) / 256);     // N of mlp_m.2
  out->cbx = out->WxP / 128;
  out->cbm = out->MP / 128;
  out->WhP = round_up(Wh, 128);
  out->HP = round_up(H, 32);
  out->K1P = round_up(H + out->MP, 8);
  out->K1Q = round_up(H + out->MP, 16);
  out->TC = 2 * out->WxP + 2 * out->WmP;
  return EGNN_OK;
}

int graph_args_check(bool have_model, int N, int E, int B, const void* edge_dst, const void* edge_src, const void* row_ptr,
                     const void* graph_ptr, const void* node_graph) {
  if (N < 1 || E < 0 || B < 1 || !row_ptr || !graph_ptr || !node_graph || (E > 0 && (!edge_dst || !edge_src))) {
    set_error("bad graph arguments");
    return EGNN_EINVAL;
  }
  if (!have_model) { set_error("egnn_set_model first"); return EGNN_ESTATE; }
  return EGNN_OK;
}

int precision_scope_check(bool ready, int prec, int norm_scope) {
  if (!ready) { set_error("model/graph not set"); return EGNN_ESTATE; }
  if (prec != EGNN_PREC_F32 && prec != EGNN_PREC_BF16 && prec != EGNN_PREC_BF16X3 && prec != EGNN_PREC_F16 && prec != EGNN_PREC_F16C8) {
    set_error("bad precision %d", prec);
    return EGNN_EINVAL;
  }
  if (norm_scope != EGNN_NORM_CALL && norm_scope != EGNN_NORM_GRAPH) { set_error("bad norm scope"); return EGNN_EINVAL; }
  return EGNN_OK;
}

void plan_gemm_tn(int E, int M, int N, int& BN, int& tiles_n, int& S, int& steps_per_slice) {
  BN = (N % 256 == 0) ? 256 : 128;
  tiles_n = N / BN;
  const int ntiles = (M / kGemmBM) * tiles_n;
  const int total_steps = (E + kGemmBK - 1) / kGemmBK;
  int want = (768 + ntiles - 1) / ntiles;                    // ~3 workgroups per CU in all
  const int max_s = total_steps / 16 > 0 ? total_steps / 16 : 1;   // at least 16 steps per slice
  S = want < 1 ? 1 : (want > max_s ? max_s : want);
  steps_per_slice = (total_steps + S - 1) / S;
  S = (total_steps + steps_per_slice - 1) / steps_per_slice;
}

static bool gemm_tn_shape_ok(int E, int M, int N) { return !(E < 1 || M < 256 || M % 256 != 0 || N < 128 || N % 128 != 0); }

int gemm_tn_args_check(int E, int M, int N, const void* A, int lda, const void* B, int ldb, const void* C, int ldc, int rows,
                       int cols, const void* workspace, size_t workspace_bytes) {
  if (!gemm_tn_shape_ok(E, M, N) || lda < M || ldb < N || lda % 8 != 0 || ldb % 8 != 0 || !A || !B || !C || rows < 1 || rows > M ||
      cols < 1 || cols > N || ldc < cols) {
    set_error("egnn_gemm_tn_bf16: unsupported shape E=%d M=%d N=%d lda=%d ldb=%d", E, M, N, lda, ldb);
    return EGNN_EINVAL;
  }
  // The kernel forms 32-bit byte offsets for the rows of its LAST k-step too (up to kGemmBK - 1 rows past E, answered with
  // zeros by the buffer range check): those offsets must not wrap either.
  const size_t lim = (size_t)1 << 32;
  if ((size_t)(E + kGemmBK - 1) * lda * 2 >= lim || (size_t)(E + kGemmBK - 1) * ldb * 2 >= lim) {
    set_error("egnn_gemm_tn_bf16: operand larger than 4 GiB (cut the reduction into chunks)");
    return EGNN_EINVAL;
  }
  int BN, tiles_n, S, sps;
  plan_gemm_tn(E, M, N, BN, tiles_n, S, sps);
  if (!workspace || workspace_bytes < (size_t)S * M * N * sizeof(float)) {
    set_error("egnn_gemm_tn_bf16: workspace too small (%zu bytes, see egnn_gemm_tn_workspace_bytes)", workspace_bytes);
    return EGNN_EINVAL;
  }
  return EGNN_OK;
}

int gemm_rows_args_check(int E, const void* A0, int lda0, int K0, const void* W0, const void* A1, int lda1, int K1, const void* W1,
                         const void* out, int ldo) {
  if (E < 1 || !A0 || !W0 || !out || K0 < 64 || K0 % 64 != 0 || lda0 < K0 || lda0 % 8 != 0 || ldo < 128 || ldo % 4 != 0 ||
      (A1 && (!W1 || K1 < 64 || K1 % 64 != 0 || lda1 < K1 || lda1 % 8 != 0))) {
    set_error("egnn_gemm_rows_bf16: unsupported shape E=%d K0=%d K1=%d", E, K0, K1);
    return EGNN_EINVAL;
  }
  // a workgroup covers 256 rows: the offsets of the rows past E in the last workgroup must not wrap (see gemm_tn_args_check)
  const size_t lim = (size_t)1 << 32;
  if ((size_t)(E + 255) * lda0 * 2 >= lim || (A1 && (size_t)(E + 255) * lda1 * 2 >= lim)) {
    set_error("egnn_gemm_rows_bf16: operand larger than 4 GiB (cut the rows into chunks)");
    return EGNN_EINVAL;
  }
  return EGNN_OK;
}

int dense_rows_args_check(int N, int K, int J, const void* in, const void* W, const void* b, const void* out) {
  if (N < 1 || K < 1 || J < 1 || K > 2048 || !in || !W || !b || !out) {
    set_error("bad egnn_dense_rows arguments (K <= 2048)");
    return EGNN_EINVAL;
  }
  return EGNN_OK;
}

bool fork_candidate(int E, int WxP) {
  const long tiles = ((long)E + 127) / 128, xw = tiles * (WxP >= 512 ? WxP / 512 : 1);
  const long idle = (xw + 255) / 256 * 256 - xw;
  return E > 0 && idle >= (tiles + 1) / 2;
}

}  // namespace egnn

using namespace egnn;

extern "C" {

const char* egnn_last_error(void) { return g_err; }
int egnn_version(void) { return 1; }

size_t egnn_gemm_tn_workspace_bytes(int E, int M, int N) {   // 0 = unsupported shape
  if (!gemm_tn_shape_ok(E, M, N)) return 0;
  int BN, tiles_n, S, sps;
  plan_gemm_tn(E, M, N, BN, tiles_n, S, sps);
  return (size_t)S * M * N * sizeof(float);
}

// ---- schedule (host, fp32, same operation order as torch on CPU) ----------------------------------
int schedule_table_from_alpha(int T, const float* alpha, const float* sigma, float* table) {
  if (T < 1 || !alpha || !table) { set_error("bad schedule arguments"); return EGNN_EINVAL; }
  for (int t = 1; t <= T; ++t) {
    // calculate_mu / reverse_diffuse_one_step, diffusion_x_h.py:61-90 (fp32 scalar arithmetic)
    const float at = alpha[t], as = alpha[t - 1];
    const float sq_t = 1.0f - at * at, sq_s = 1.0f - as * as;
    const float ats = at / as;
    const float sq_ts = sq_t - (ats * ats) * sq_s;
    const float sig_t = sqrtf(sq_t);
    table[4 * t + 0] = 1.0f / ats;
    table[4 * t + 1] = sq_ts / ats / sig_t;
    table[4 * t + 2] = sqrtf(sq_ts * sq_s / sq_t);
    table[4 * t + 3] = (float)t / (float)T;
  }
  const float a0 = alpha[0], s0 = sigma ? sigma[0] : sqrtf(1.0f - a0 * a0);
  table[0] = 1.0f / a0; table[1] = s0 / a0; table[2] = s0 / a0; table[3] = 0.f;
  return EGNN_OK;
}

int schedule_table_build(int T, double s, double power, float* alpha, float* sigma, float* table) {
  if (T < 1) { set_error("T must be >= 1"); return EGNN_EINVAL; }
  std::vector<float> a(T + 1), sg(T + 1);
  // polynomial_schedule (:99-106): x = linspace(0,T,T+1); a2 = (1 - (x/T)^p)^2
  // clip_noise_schedule (:92-97): ratios to the previous entry (first vs 1), clamp [0.001, 1], cumprod.
  // Python scalars enter torch's fp32 tensor arithmetic rounded to fp32: (1 - 2*s) and s are formed in
  // double first; torch.pow with exponent 2 / 3 is evaluated as x*x / x*x*x.
  const float prec = (float)(1.0 - 2.0 * s), sf = (float)s, pw = (float)power;
  // torch.cumprod on CPU accumulates fp32 inputs in double and rounds each output to fp32
  float prev = 1.0f;
  double cum = 1.0;
  for (int i = 0; i <= T; ++i) {
    const float q = (float)i / (float)T;  // linspace(0, T, T+1) is exact for integer endpoints
    float qp;
    if (power == 2.0) qp = q * q;
    else if (power == 3.0) qp = q * q * q;
    else if (power == 1.0) qp = q;
    else qp = powf(q, pw);
    const float base = 1.0f - qp;
    const float a2 = base * base;
    float step = a2 / prev;
    step = fminf(fmaxf(step, 0.001f), 1.0f);
    cum = (i == 0) ? (double)step : cum * (double)step;
    prev = a2;
    a[i] = prec * (float)cum + sf;
    sg[i] = sqrtf(1.0f - a[i] * a[i]);
  }
  if (alpha) memcpy(alpha, a.data(), sizeof(float) * (T + 1));
  if (sigma) memcpy(sigma, sg.data(), sizeof(float) * (T + 1));
  if (table) return schedule_table_from_alpha(T, a.data(), sg.data(), table);
  return EGNN_OK;
}

// Entry points of the CPU-only sanitizer build (make asan): the validation layer of the device entry points, callable
// without a GPU.  Not exported by libegnn_amd.so (compiled only with -DEGNN_HOST_TEST_API).
#ifdef EGNN_HOST_TEST_API
int egnn_host_model_dims(int L, int H, int M, int Wm, int Wx, int Wh, int* out10) {
  ModelDims d;
  const int rc = model_dims(L, H, M, Wm, Wx, Wh, &d);
  if (rc == EGNN_OK && out10) {
    const int v[10] = {d.WxP, d.WmP, d.MP, d.WhP, d.HP, d.K1P, d.K1Q, d.TC, d.cbx, d.cbm};
    memcpy(out10, v, sizeof(v));
  }
  return rc;
}
int egnn_host_graph_args_check(int have_model, int N, int E, int B, const void* a, const void* b, const void* c, const void* d,
                               const void* e) {
  return graph_args_check(have_model != 0, N, E, B, a, b, c, d, e);
}
int egnn_host_precision_scope_check(int ready, int prec, int scope) { return precision_scope_check(ready != 0, prec, scope); }
int egnn_host_plan_gemm_tn(int E, int M, int N, int* out4) {
  int BN, tn, S, sps;
  plan_gemm_tn(E, M, N, BN, tn, S, sps);
  out4[0] = BN; out4[1] = tn; out4[2] = S; out4[3] = sps;
  return EGNN_OK;
}
int egnn_host_gemm_tn_args_check(int E, int M, int N, const void* A, int lda, const void* B, int ldb, const void* C, int ldc, int rows,
                                 int cols, const void* ws, size_t wsb) {
  return gemm_tn_args_check(E, M, N, A, lda, B, ldb, C, ldc, rows, cols, ws, wsb);
}
int egnn_host_gemm_rows_args_check(int E, const void* A0, int lda0, int K0, const void* W0, const void* A1, int lda1, int K1,
                                   const void* W1, const void* out, int ldo) {
  return gemm_rows_args_check(E, A0, lda0, K0, W0, A1, lda1, K1, W1, out, ldo);
}
int egnn_host_dense_rows_args_check(int N, int K, int J, const void* in, const void* W, const void* b, const void* out) {
  return dense_rows_args_check(N, K, J, in, W, b, out);
}
int egnn_host_fork_candidate(int E, int WxP) { return fork_candidate(E, WxP) ? 1 : 0; }
#endif

}  // extern "C"
