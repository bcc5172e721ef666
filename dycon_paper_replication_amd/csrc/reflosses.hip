// The reference's loss callables with the reference's OWN argument meaning (code/utils/losses.py:8-16, 65-104, 156-192):
// probabilities (or any float tensor) in, element-wise tensor / scalar out.  The fused step (trainer.py) does not use these -- it
// reads the logits once (losses.hip) -- they exist so that the reference's training-loop body runs unchanged on this package
// (INTEGRATION.md section 1).  All are streaming kernels over a strided (n, C, V) view: channel stride 1 for the package's
// channels-last tensors, channel stride V for a plain NCDHW tensor, element stride 2 for `probs[:, 1]` of a 2-class map.
#include "common.h"

constexpr int MAXC = 8;

struct View { const float* p; long long sn, sc, sv; };
static inline View mkview(const dycon_view_t* v) { return View{(const float*)v->p, v->sn, v->sc, v->sv}; }

template <int DUMMY = 0>
__device__ __forceinline__ void load_probs(const View& a, long long n, long long v, int C, int sigmoid, float* p, float* lp) {
    // p = softmax over the C channels (or element-wise sigmoid); lp = log p
    const float* base = a.p + n * a.sn + v * a.sv;
    float x[MAXC], m = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C) { x[c] = base[c * a.sc]; m = fmaxf(m, x[c]); }
    if (sigmoid) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) { p[c] = 1.f / (1.f + expf(-x[c])); lp[c] = logf(p[c]); }
        return;
    }
    float z = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C) { p[c] = expf(x[c] - m); z += p[c]; }
    const float lz = logf(z);
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C) { lp[c] = x[c] - m - lz; p[c] = p[c] / z; }
}

// ------------------------------------------------------------------ softmax_mse_loss (losses.py:65-82): element-wise (p - q)^2
__global__ __launch_bounds__(256) void softmax_mse_fwd_kernel(View a, View b, float* __restrict__ out, long long osn, long long osc,
                                                              long long osv, long long n_, int C, long long V, int sigmoid) {
    const long long total = n_ * V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / V, v = i - n * V;
        float p[MAXC], q[MAXC], lp[MAXC];
        load_probs(a, n, v, C, sigmoid, p, lp);
        load_probs(b, n, v, C, sigmoid, q, lp);
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) { const float d = p[c] - q[c]; out[n * osn + c * osc + v * osv] = d * d; }
    }
}

// gradient w.r.t. the FIRST argument (the loss is symmetric: call with (b, a) for the second)
__global__ __launch_bounds__(256) void softmax_mse_bwd_kernel(View a, View b, View g, float* __restrict__ ga, long long osn,
                                                              long long osc, long long osv, long long n_, int C, long long V,
                                                              int sigmoid) {
    const long long total = n_ * V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / V, v = i - n * V;
        float p[MAXC], q[MAXC], lp[MAXC], h[MAXC];
        load_probs(a, n, v, C, sigmoid, p, lp);
        load_probs(b, n, v, C, sigmoid, q, lp);
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) { h[c] = 2.f * g.p[n * g.sn + c * g.sc + v * g.sv] * (p[c] - q[c]); dot += h[c] * p[c]; }
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) ga[n * osn + c * osc + v * osv] = sigmoid ? h[c] * p[c] * (1.f - p[c]) : p[c] * (h[c] - dot);
    }
}

// ------------------------------------------------------------------ softmax_kl_loss (losses.py:85-104): F.kl_div(log p, q, 'mean')
__device__ __forceinline__ void block_atomic_double(float v, double* dst) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum(v);
    if (lane == 0) red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dst, (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]);
    __syncthreads();
}

__global__ __launch_bounds__(256) void softmax_kl_fwd_kernel(View a, View b, long long n_, int C, long long V, int sigmoid,
                                                             double* __restrict__ sum) {
    const long long total = n_ * V;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / V, v = i - n * V;
        float p[MAXC], q[MAXC], lp[MAXC], lq[MAXC];
        load_probs(a, n, v, C, sigmoid, p, lp);
        load_probs(b, n, v, C, sigmoid, q, lq);
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) acc += q[c] > 0.f ? q[c] * (lq[c] - lp[c]) : 0.f;     // xlogy: 0 where the target is 0
    }
    block_atomic_double(acc, sum);
}

__global__ void scalar_mean_kernel(const double* __restrict__ sum, double count, float* __restrict__ out) {
    out[0] = (float)(sum[0] / count);
}

// which = 0: d/d input logits = g/count * (p * sum(q) - q)   (sigmoid: -q (1 - p));
// which = 1: d/d target logits = g/count * q (r - sum q r), r = log q - log p + 1   (sigmoid: q (1-q) r)
__global__ __launch_bounds__(256) void softmax_kl_bwd_kernel(View a, View b, long long n_, int C, long long V, int sigmoid,
                                                             int which, const float* __restrict__ g_up, double count,
                                                             float* __restrict__ gr, long long osn, long long osc, long long osv) {
    const long long total = n_ * V;
    const float s = g_up[0] / (float)count;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / V, v = i - n * V;
        float p[MAXC], q[MAXC], lp[MAXC], lq[MAXC];
        load_probs(a, n, v, C, sigmoid, p, lp);
        load_probs(b, n, v, C, sigmoid, q, lq);
        float qs = 0.f, qr = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) { qs += q[c]; qr += q[c] * (lq[c] - lp[c] + 1.f); }
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                float r;
                if (which == 0) r = sigmoid ? -q[c] * (1.f - p[c]) : p[c] * qs - q[c];
                else {
                    const float rc = lq[c] - lp[c] + 1.f;
                    r = sigmoid ? q[c] * (1.f - q[c]) * rc : q[c] * (rc - qr);
                }
                gr[n * osn + c * osc + v * osv] = s * r;
            }
    }
}

// ------------------------------------------------------------------ dice_loss / DiceLoss (losses.py:8-16, 156-192)
// target kinds: 0 float32, 1 one byte (uint8 / bool), 2 int64.  onehot = 1: the target is a label map (n, V) and class c's target is
// (label == c) (DiceLoss._one_hot_encoder); onehot = 0: the target has the score's logical shape and is used as is (dice_loss).
__device__ __forceinline__ float load_target(const void* t, int kind, long long off) {
    return kind == 0 ? ((const float*)t)[off] : kind == 1 ? (float)((const uint8_t*)t)[off] : (float)((const long long*)t)[off];
}

__device__ __forceinline__ void load_scores(const View& a, long long n, long long v, int C, int softmax, float* s) {
    if (softmax) {
        float lp[MAXC];
        load_probs(a, n, v, C, 0, s, lp);
    } else {
        const float* base = a.p + n * a.sn + v * a.sv;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) s[c] = base[c * a.sc];
    }
}

__global__ __launch_bounds__(256) void dice_sums_kernel(View a, const void* __restrict__ tgt, int tkind, int onehot, long long tsn,
                                                        long long tsc, long long tsv, long long n_, int C, long long V, int softmax,
                                                        double* __restrict__ sums) {
    __shared__ float red[4][MAXC * 3];
    float acc[MAXC * 3];
#pragma unroll
    for (int k = 0; k < MAXC * 3; ++k) acc[k] = 0.f;
    const long long total = n_ * V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / V, v = i - n * V;
        float s[MAXC];
        load_scores(a, n, v, C, softmax, s);
        const float lab = onehot ? load_target(tgt, tkind, n * tsn + v * tsv) : 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                const float t = onehot ? (lab == (float)c ? 1.f : 0.f) : load_target(tgt, tkind, n * tsn + c * tsc + v * tsv);
                acc[3 * c] += s[c] * t; acc[3 * c + 1] += s[c] * s[c]; acc[3 * c + 2] += t * t;
            }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < MAXC * 3; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) red[w][k] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < 3 * C) {
        const int k = threadIdx.x;
        atomicAdd(&sums[k], (double)red[0][k] + (double)red[1][k] + (double)red[2][k] + (double)red[3][k]);
    }
}

struct Weights { float w[MAXC]; };

// loss = sum_c w_c (1 - (2 I_c + smooth) / (Z_c + Y_c + smooth)) / n_div
__global__ void dice_finalize_kernel(const double* __restrict__ sums, int C, Weights wt, float n_div, float* __restrict__ out) {
    const float smooth = 1e-5f;
    float loss = 0.f;
    for (int c = 0; c < C; ++c) {
        const float I = (float)sums[3 * c], Z = (float)sums[3 * c + 1], Y = (float)sums[3 * c + 2];
        loss += wt.w[c] * (1.f - (2.f * I + smooth) / (Z + Y + smooth));
    }
    out[0] = loss / n_div;
}

__global__ __launch_bounds__(256) void dice_bwd_kernel(View a, const void* __restrict__ tgt, int tkind, int onehot, long long tsn,
                                                       long long tsc, long long tsv, long long n_, int C, long long V, int softmax,
                                                       const double* __restrict__ sums, const float* __restrict__ g_up, Weights wt,
                                                       float n_div, float* __restrict__ gr, long long osn, long long osc,
                                                       long long osv) {
    const float smooth = 1e-5f;
    float I2[MAXC], D[MAXC], k[MAXC];
    const float gu = g_up[0] / n_div;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C) {
            I2[c] = 2.f * (float)sums[3 * c] + smooth;
            D[c] = (float)sums[3 * c + 1] + (float)sums[3 * c + 2] + smooth;
            k[c] = gu * wt.w[c];
        }
    const long long total = n_ * V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / V, v = i - n * V;
        float s[MAXC], h[MAXC];
        load_scores(a, n, v, C, softmax, s);
        const float lab = onehot ? load_target(tgt, tkind, n * tsn + v * tsv) : 0.f;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                const float t = onehot ? (lab == (float)c ? 1.f : 0.f) : load_target(tgt, tkind, n * tsn + c * tsc + v * tsv);
                h[c] = -k[c] * (2.f * t * D[c] - I2[c] * 2.f * s[c]) / (D[c] * D[c]);
                dot += h[c] * s[c];
            }
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) gr[n * osn + c * osc + v * osv] = softmax ? s[c] * (h[c] - dot) : h[c];
    }
}

// ------------------------------------------------------------------ host entry points
static inline int grid_for(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
#define CHECK_VIEW(v) DYCON_REQUIRE((v) && (v)->p, "null view")
#define CHECK_DIMS() DYCON_REQUIRE(n >= 0 && V >= 0 && C >= 1 && C <= MAXC, "bad dims n=%lld C=%d V=%lld (1 <= C <= %d)", n, C, V, MAXC)

extern "C" int dycon_softmax_mse_fwd(const dycon_view_t* a, const dycon_view_t* b, const dycon_view_t* out, long long n, int C,
                                     long long V, int sigmoid, dycon_stream_t stream) {
    CHECK_VIEW(a); CHECK_VIEW(b); CHECK_VIEW(out); CHECK_DIMS();
    if (n * V == 0) return DYCON_OK;
    softmax_mse_fwd_kernel<<<grid_for(n * V), 256, 0, stream>>>(mkview(a), mkview(b), (float*)out->p, out->sn, out->sc, out->sv, n, C, V,
                                                                sigmoid);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_softmax_mse_bwd(const dycon_view_t* a, const dycon_view_t* b, const dycon_view_t* g, const dycon_view_t* ga,
                                     long long n, int C, long long V, int sigmoid, dycon_stream_t stream) {
    CHECK_VIEW(a); CHECK_VIEW(b); CHECK_VIEW(g); CHECK_VIEW(ga); CHECK_DIMS();
    if (n * V == 0) return DYCON_OK;
    softmax_mse_bwd_kernel<<<grid_for(n * V), 256, 0, stream>>>(mkview(a), mkview(b), mkview(g), (float*)ga->p, ga->sn, ga->sc, ga->sv,
                                                                n, C, V, sigmoid);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_softmax_kl_fwd(const dycon_view_t* a, const dycon_view_t* b, long long n, int C, long long V, int sigmoid,
                                    double* sum, float* out, dycon_stream_t stream) {
    CHECK_VIEW(a); CHECK_VIEW(b); CHECK_DIMS();
    DYCON_REQUIRE(sum && out && n * V > 0, "softmax_kl: empty input or null output");
    if (hipMemsetAsync(sum, 0, sizeof(double), stream) != hipSuccess) { dycon_set_error("memset failed"); return DYCON_ERR_LAUNCH; }
    softmax_kl_fwd_kernel<<<grid_for(n * V), 256, 0, stream>>>(mkview(a), mkview(b), n, C, V, sigmoid, sum);
    scalar_mean_kernel<<<1, 1, 0, stream>>>(sum, (double)n * C * V, out);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_softmax_kl_bwd(const dycon_view_t* a, const dycon_view_t* b, long long n, int C, long long V, int sigmoid,
                                    int which, const float* g_up, const dycon_view_t* grad, dycon_stream_t stream) {
    CHECK_VIEW(a); CHECK_VIEW(b); CHECK_VIEW(grad); CHECK_DIMS();
    DYCON_REQUIRE(g_up && (which == 0 || which == 1), "softmax_kl_bwd: bad arguments");
    if (n * V == 0) return DYCON_OK;
    softmax_kl_bwd_kernel<<<grid_for(n * V), 256, 0, stream>>>(mkview(a), mkview(b), n, C, V, sigmoid, which, g_up, (double)n * C * V,
                                                               (float*)grad->p, grad->sn, grad->sc, grad->sv);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

static inline Weights mkweights(const float* w_host, int C) {
    Weights wt;
    for (int c = 0; c < MAXC; ++c) wt.w[c] = (w_host && c < C) ? w_host[c] : 1.f;
    return wt;
}

extern "C" int dycon_dice_fwd(const dycon_view_t* score, const dycon_view_t* target, int target_kind, int onehot, long long n, int C,
                              long long V, int softmax, const float* weights_host, float n_div, double* sums, float* out,
                              dycon_stream_t stream) {
    CHECK_VIEW(score); CHECK_VIEW(target); CHECK_DIMS();
    DYCON_REQUIRE(sums && out && target_kind >= 0 && target_kind <= 2, "dice_fwd: bad arguments");
    if (hipMemsetAsync(sums, 0, sizeof(double) * 3 * MAXC, stream) != hipSuccess) { dycon_set_error("memset failed"); return DYCON_ERR_LAUNCH; }
    if (n * V > 0)
        dice_sums_kernel<<<grid_for(n * V), 256, 0, stream>>>(mkview(score), target->p, target_kind, onehot, target->sn, target->sc,
                                                              target->sv, n, C, V, softmax, sums);
    dice_finalize_kernel<<<1, 1, 0, stream>>>(sums, C, mkweights(weights_host, C), n_div, out);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_dice_bwd(const dycon_view_t* score, const dycon_view_t* target, int target_kind, int onehot, long long n, int C,
                              long long V, int softmax, const float* weights_host, float n_div, const double* sums, const float* g_up,
                              const dycon_view_t* grad, dycon_stream_t stream) {
    CHECK_VIEW(score); CHECK_VIEW(target); CHECK_VIEW(grad); CHECK_DIMS();
    DYCON_REQUIRE(sums && g_up && target_kind >= 0 && target_kind <= 2, "dice_bwd: bad arguments");
    if (n * V == 0) return DYCON_OK;
    dice_bwd_kernel<<<grid_for(n * V), 256, 0, stream>>>(mkview(score), target->p, target_kind, onehot, target->sn, target->sc, target->sv,
                                                         n, C, V, softmax, sums, g_up, mkweights(weights_host, C), n_div,
                                                         (float*)grad->p, grad->sn, grad->sc, grad->sv);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}
