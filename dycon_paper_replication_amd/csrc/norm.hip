// GroupNorm / InstanceNorm / train-mode BatchNorm on NDHWC activations, one kernel family.
//
//   x: (Nb, V, C), G groups of cpg = C/G channels; statistics over V*cpg elements per (n, g).
//   GroupNorm(16, C)  : VNet.py:20            -> G = 16, affine
//   InstanceNorm3d(C) : networks/utils.py:105 -> G = C, no affine
//   BatchNorm3d(C)    : UNet3D_contrastive.py:263,266 (train mode) -> Nb = 1, V = all B*N rows, G = C
//
// All passes are pure HBM streams (16-byte loads, channel-fixed per thread so the per-channel
// partials live in registers); the grid-wide reduction is two-stage and deterministic:
// per-chunk partials -> tiny finalize kernel (double accumulation).
#include "common.h"
#include <type_traits>

struct NormPlan { int chunks; long long rows_per_chunk; };
static NormPlan norm_plan(long long V) {
    NormPlan p;
    static const long long maxc = [] { const char* v = getenv("DYCON_NORM_CHUNKS"); return v && *v ? atoll(v) : 256LL; }();
    long long rpc = (V + maxc - 1) / maxc;     // <= 256 chunks per sample ...
    static const long long minr = [] { const char* v = getenv("DYCON_NORM_MIN_ROWS"); return v && *v ? atoll(v) : 128LL; }();
    if (rpc < minr) rpc = minr;          // ... of >= 128 rows.  (Round 3: 512 chunks of >= 64 rows -> 256 of >= 128: the step 4.54 -> 4.47 ms over two
                                         // pairs, profiles/r03_norm_chunk_plan.txt -- per-workgroup fixed cost (launch ramp, the LDS tree, the partial
                                         // stores and the finalize that sums them) against the length of the serial row loop; 128 / 1024 chunks are slower.)
    p.rows_per_chunk = rpc;
    p.chunks = (int)((V + rpc - 1) / rpc);
    return p;
}

// ---- stage 1: per-(n, chunk, channel) sums.  MODE 0: {sum x, sum x^2}; MODE 1: {sum g, sum g*xhat}
// acc != NULL ("accumulator form"): instead of storing the chunk's partials for a finalize launch, every chunk adds its two sums per
// channel to acc[n][c][2] (double atomics: the order of the <= 512 adds per address changes the last bits of a double, far below
// the float the consumer rounds to); the consumer kernel forms the group statistics in its prologue from those C x 2 doubles.
// One dependent launch less per normalisation, forward and backward, on the step's critical chain.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void norm_partial_kernel(const T* __restrict__ S, const T* __restrict__ GY,
                                                           float* __restrict__ part, long long V, int C, int G,
                                                           long long rows_per_chunk, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int relu, int from_y, const float* __restrict__ chan_scale,
                                                           double* __restrict__ acc = nullptr, int nslots = 1) {
    constexpr int VN = Vec16<T>::N;
    // fp32 storage (the 1e-4 parity mode): the per-thread sums are kept in double -- that mode's budget against the fp64 twin of the
    // reference (twice the reference's own fp32 error on the logits after an optimiser step) leaves no room for an fp32 chain whose
    // length depends on the chunk plan.  bf16 storage: fp32 (the stored values carry 2^-9 noise each).
    typedef typename std::conditional<sizeof(T) == 4, double, float>::type A;
    __shared__ A sm[2][256][VN + 1];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int ngrp = C / VN;            // 16-byte groups per row
    const int rpi = 256 / ngrp;         // rows per iteration (0 if C/VN > 256: rejected on the host)
    const int cg = threadIdx.x % ngrp, rr = threadIdx.x / ngrp;
    const long long v0 = chunk * rows_per_chunk, v1 = min(V, v0 + rows_per_chunk);
    const int cpg = C / G;
    A a0[VN], a1[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) a0[k] = a1[k] = 0;
    float mean[VN], rstd[VN], gm[VN], bt[VN], cs[VN];
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            const int c = cg * VN + k;
            const int g = c / cpg;
            cs[k] = chan_scale ? chan_scale[(long long)n * C + c] : 1.f;
            mean[k] = stats[((long long)n * G + g) * 2];
            rstd[k] = stats[((long long)n * G + g) * 2 + 1];
            gm[k] = gamma ? gamma[c] : 1.f;
            bt[k] = beta ? beta[c] : 0.f;
        }
    }
    if (rr < rpi) {
        const T* sp = S + ((long long)n * V) * C + cg * VN;
        const T* gp = MODE == 1 ? GY + ((long long)n * V) * C + cg * VN : nullptr;
        for (long long v = v0 + rr; v < v1; v += rpi) {
            const Vec16<T> s = ld16(sp + v * C);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < VN; ++k) {
                    const A x = s.get(k);
                    a0[k] += x;
                    a1[k] += x * x;
                }
            } else {
                const Vec16<T> gv = ld16(gp + v * C);
#pragma unroll
                for (int k = 0; k < VN; ++k) {
                    float xh, g = gv.get(k) * cs[k];
                    if (from_y) {
                        const float y = s.get(k);
                        xh = (y - bt[k]) / gm[k];
                        if (relu && !(y > 0.f)) g = 0.f;
                    } else {
                        xh = (s.get(k) - mean[k]) * rstd[k];
                        if (relu && !(gm[k] * xh + bt[k] > 0.f)) g = 0.f;
                    }
                    a0[k] += (A)g;
                    a1[k] += (A)g * (A)xh;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < VN; ++k) { sm[0][threadIdx.x][k] = a0[k]; sm[1][threadIdx.x][k] = a1[k]; }
    __syncthreads();
    // thread (cg, k) pairs: C outputs x 2
    for (int o = threadIdx.x; o < C; o += 256) {
        const int og = o / VN, ok = o % VN;
        // the workgroup's per-thread sums are combined in double: with 256 chunks per sample (round 3; 512 before) a chunk's sum is a
        // longer fp32 chain, and the fp32 parity mode's logits budget (twice the reference's own fp32 error) is tight at step 1
        double s0d = 0.0, s1d = 0.0;
        for (int q = 0; q < rpi; ++q) { s0d += (double)sm[0][q * ngrp + og][ok]; s1d += (double)sm[1][q * ngrp + og][ok]; }
        const float s0 = (float)s0d, s1 = (float)s1d;
        if (acc) {     // chunk -> one of nslots copies: <= chunks/nslots adds per address (same-address atomics serialise at the memory side)
            double* a = acc + (((long long)n * nslots + chunk % nslots) * C + o) * 2;
            atomicAdd(a, (double)s0);
            atomicAdd(a + 1, (double)s1);
            continue;
        }
        float* d = part + ((((long long)n * gridDim.x + chunk) * C) + o) * 2;
        d[0] = s0;
        d[1] = s1;
    }
}

// two doubles summed over the 256 threads of a workgroup (result valid in thread 0)
__device__ __forceinline__ void block_sum2_double(double& s0, double& s1) {
    __shared__ double red[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = s0; red[1][w] = s1; }
    __syncthreads();
    s0 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    s1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
}

// ---- stage 2 (forward): mean / rstd per (n, g) (+ BatchNorm running statistics).  One 256-thread workgroup per (n, g): the <= 512 x cpg
// partials are two loads per thread, all in flight at once (this launch sits on the step's dependent chain 38 times).
__global__ __launch_bounds__(256) void norm_finalize_stats_kernel(const float* __restrict__ part, int Nb, int chunks, int C, int G,
                                                                  long long V, float eps, float* __restrict__ stats,
                                                                  float* running_mean, float* running_var, float momentum) {
    const int i = blockIdx.x;
    const int n = i / G, g = i % G, cpg = C / G;
    double s0 = 0.0, s1 = 0.0;
    const int items = chunks * cpg;
    for (int e = threadIdx.x; e < items; e += 256) {
        const int ch = e / cpg, c = g * cpg + e % cpg;
        const float* p = part + ((((long long)n * chunks + ch) * C) + c) * 2;
        s0 += p[0];
        s1 += p[1];
    }
    block_sum2_double(s0, s1);
    if (threadIdx.x != 0) return;
    const double cnt = (double)V * cpg;
    const double mean = s0 / cnt;
    double var = s1 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(long long)i * 2] = (float)mean;
    stats[(long long)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean && Nb == 1 && G == C) {   // nn.BatchNorm3d: unbiased variance in the running estimate
        running_mean[g] = (1.f - momentum) * running_mean[g] + momentum * (float)mean;
        const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
        running_var[g] = (1.f - momentum) * running_var[g] + momentum * (float)unb;
    }
}

// ---- stage 2 (backward): blocks [0, Nb*G): per-(n,g) {A, B}; blocks [Nb*G, Nb*G + C): per-channel dgamma/dbeta
__global__ __launch_bounds__(256) void norm_finalize_bwd_kernel(const float* __restrict__ part, int Nb, int chunks, int C, int G,
                                                                const float* __restrict__ gamma, float* __restrict__ ab,
                                                                float* dgamma, float* dbeta) {
    const int cpg = C / G;
    double s0 = 0.0, s1 = 0.0;
    if ((int)blockIdx.x < Nb * G) {
        const int i = blockIdx.x, n = i / G, g = i % G;
        const int items = chunks * cpg;
        for (int e = threadIdx.x; e < items; e += 256) {
            const int ch = e / cpg, c = g * cpg + e % cpg;
            const float* p = part + ((((long long)n * chunks + ch) * C) + c) * 2;
            const double gm = gamma ? (double)gamma[c] : 1.0;
            s0 += gm * p[0];
            s1 += gm * p[1];
        }
        block_sum2_double(s0, s1);
        if (threadIdx.x == 0) { ab[(long long)i * 2] = (float)s0; ab[(long long)i * 2 + 1] = (float)s1; }
    } else {
        const int c = blockIdx.x - Nb * G;
        const int items = Nb * chunks;
        for (int e = threadIdx.x; e < items; e += 256) {
            const float* p = part + ((long long)e * C + c) * 2;   // (n, chunk) pairs are contiguous blocks of C
            s0 += p[0];
            s1 += p[1];
        }
        block_sum2_double(s0, s1);
        if (threadIdx.x == 0) { if (dbeta) dbeta[c] = (float)s0; if (dgamma) dgamma[c] = (float)s1; }
    }
}

// ---- apply: y = act(x*scale[c] + shift[c]) + skip      (per sample n = blockIdx.y)
// ACC: the statistics are not in `stats` yet -- form them from the per-channel {sum x, sum x^2} accumulators of norm_partial_kernel
// (accumulator form) and let workgroup 0 of each sample publish them (stats_out, BatchNorm running statistics) for the backward.
template <typename T, bool ACC>
__global__ __launch_bounds__(256) void norm_apply_kernel(const T* __restrict__ X, T* __restrict__ Y, long long V, int C, int G,
                                                         const float* __restrict__ stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int relu, const T* __restrict__ skip,
                                                         const float* __restrict__ chan_scale, const double* __restrict__ acc = nullptr,
                                                         float* __restrict__ stats_out = nullptr, float eps = 0.f,
                                                         float* running_mean = nullptr, float* running_var = nullptr,
                                                         float momentum = 0.f, int nslots = 1) {
    constexpr int VN = Vec16<T>::N;
    extern __shared__ float ss[];  // scale[C], shift[C], channel (dropout) scale[C]
    const int n = blockIdx.y, cpg = C / G;
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cpg;
        float mean, rstd;
        if (ACC) {
            double s0 = 0.0, s1 = 0.0;
            for (int sl = 0; sl < nslots; ++sl)
                for (int j = 0; j < cpg; ++j) {
                    const double* a = acc + (((long long)n * nslots + sl) * C + g * cpg + j) * 2;
                    s0 += a[0];
                    s1 += a[1];
                }
            const double cnt = (double)V * cpg, m = s0 / cnt;
            double var = s1 / cnt - m * m;
            if (var < 0.0) var = 0.0;
            mean = (float)m;
            rstd = (float)(1.0 / sqrt(var + (double)eps));
            if (blockIdx.x == 0 && c == g * cpg) {
                stats_out[((long long)n * G + g) * 2] = mean;
                stats_out[((long long)n * G + g) * 2 + 1] = rstd;
                if (running_mean && gridDim.y == 1 && G == C) {   // nn.BatchNorm3d: unbiased variance in the running estimate
                    running_mean[g] = (1.f - momentum) * running_mean[g] + momentum * mean;
                    const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
                    running_var[g] = (1.f - momentum) * running_var[g] + momentum * (float)unb;
                }
            }
        } else {
            mean = stats[((long long)n * G + g) * 2];
            rstd = stats[((long long)n * G + g) * 2 + 1];
        }
        const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        ss[c] = rstd * gm;
        ss[C + c] = bt - mean * rstd * gm;
        ss[2 * C + c] = chan_scale ? chan_scale[(long long)n * C + c] : 1.f;
    }
    __syncthreads();
    const long long total = V * C / VN;
    const long long base = (long long)n * V * C;
    // channel of a thread's first element: once per thread (a 64-bit modulo per 16-byte vector used to cost as many vector
    // instructions as the normalisation itself); it advances by the grid stride modulo C -- zero whenever C divides 256 * VN * grid
    int c0 = (int)((((long long)blockIdx.x * 256 + threadIdx.x) * VN) % C);
    const int cstep = (int)(((long long)gridDim.x * 256 * VN) % C);
    if (cstep == 0) {                          // (uniform) the thread's channels never change: scale / shift / dropout factor in registers
        float sc[VN], sh[VN], dr[VN];
#pragma unroll
        for (int k = 0; k < VN; ++k) { sc[k] = ss[c0 + k]; sh[k] = ss[C + c0 + k]; dr[k] = ss[2 * C + c0 + k]; }
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long e = i * VN;
            Vec16<T> x = ld16(X + base + e), s;
            if (skip) s = ld16(skip + base + e);
            Vec16<T> y;
#pragma unroll
            for (int k = 0; k < VN; ++k) {
                float v = x.get(k) * sc[k] + sh[k];
                if (relu) v = v < 0.f ? 0.f : v;   // NaN-propagating, like torch.relu (fmaxf would swallow a NaN)
                v *= dr[k];                        // fused nn.Dropout3d: keep[n,c] / (1-p)   (1 when absent)
                if (skip) v += s.get(k);
                y.set(k, v);
            }
            st16(Y + base + e, y);
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long e = i * VN;
        Vec16<T> x = ld16(X + base + e), s;
        if (skip) s = ld16(skip + base + e);
        Vec16<T> y;
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            float v = x.get(k) * ss[c0 + k] + ss[C + c0 + k];
            if (relu) v = v < 0.f ? 0.f : v;
            v *= ss[2 * C + c0 + k];
            if (skip) v += s.get(k);
            y.set(k, v);
        }
        st16(Y + base + e, y);
        c0 += cstep;
        if (c0 >= C) c0 -= C;
    }
}

// ---- backward apply: gx = rstd * (gamma*g - (A + xhat*B)/cnt)
// ACC: {A, B} per group come from the per-channel {sum g, sum g*xhat} accumulators (accumulator form of norm_partial_kernel),
// and workgroup (0, 0) writes dgamma / dbeta (sums over the samples) -- no finalize launch.
template <typename T, bool ACC>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const T* __restrict__ S, const T* __restrict__ GY, T* __restrict__ GX,
                                                             long long V, int C, int G, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ ab, int relu, int from_y,
                                                             const float* __restrict__ chan_scale,
                                                             const double* __restrict__ acc = nullptr, float* dgamma = nullptr,
                                                             float* dbeta = nullptr, int nslots = 1) {
    constexpr int VN = Vec16<T>::N;
    extern __shared__ float ss[];  // per channel: mean, rstd, gamma, beta, A/cnt, B/cnt, dropout scale
    const int n = blockIdx.y, cpg = C / G;
    const float inv_cnt = 1.f / ((float)V * (float)cpg);
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cpg;
        ss[c] = stats[((long long)n * G + g) * 2];
        ss[C + c] = stats[((long long)n * G + g) * 2 + 1];
        ss[2 * C + c] = gamma ? gamma[c] : 1.f;
        ss[3 * C + c] = beta ? beta[c] : 0.f;
        if (ACC) {
            double a = 0.0, b = 0.0;
            for (int sl = 0; sl < nslots; ++sl)
                for (int j = 0; j < cpg; ++j) {
                    const int cj = g * cpg + j;
                    const double gm = gamma ? (double)gamma[cj] : 1.0;
                    const double* q = acc + (((long long)n * nslots + sl) * C + cj) * 2;
                    a += gm * q[0];
                    b += gm * q[1];
                }
            ss[4 * C + c] = (float)a * inv_cnt;
            ss[5 * C + c] = (float)b * inv_cnt;
            if (blockIdx.x == 0 && n == 0 && (dgamma || dbeta)) {
                double s0 = 0.0, s1 = 0.0;
                for (int m = 0; m < (int)gridDim.y * nslots; ++m) { s0 += acc[((long long)m * C + c) * 2]; s1 += acc[((long long)m * C + c) * 2 + 1]; }
                if (dbeta) dbeta[c] = (float)s0;
                if (dgamma) dgamma[c] = (float)s1;
            }
        } else {
            ss[4 * C + c] = ab[((long long)n * G + g) * 2] * inv_cnt;
            ss[5 * C + c] = ab[((long long)n * G + g) * 2 + 1] * inv_cnt;
        }
        ss[6 * C + c] = chan_scale ? chan_scale[(long long)n * C + c] : 1.f;
    }
    __syncthreads();
    const long long total = V * C / VN;
    const long long base = (long long)n * V * C;
    int c0 = (int)((((long long)blockIdx.x * 256 + threadIdx.x) * VN) % C);     // see norm_apply_kernel
    const int cstep = (int)(((long long)gridDim.x * 256 * VN) % C);
    if (cstep == 0 && !from_y) {               // (uniform) per-channel constants in registers, folded: gx = g*p + (x*q + r)
        float mu[VN], rs[VN], gmv[VN], btv[VN], pa[VN], pb[VN], dr[VN];
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            const int c = c0 + k;
            mu[k] = ss[c]; rs[k] = ss[C + c]; gmv[k] = ss[2 * C + c]; btv[k] = ss[3 * C + c];
            pa[k] = ss[4 * C + c]; pb[k] = ss[5 * C + c]; dr[k] = ss[6 * C + c];
        }
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long e = i * VN;
            const Vec16<T> s = ld16(S + base + e), gv = ld16(GY + base + e);
            Vec16<T> o;
#pragma unroll
            for (int k = 0; k < VN; ++k) {
                const float xh = (s.get(k) - mu[k]) * rs[k];
                float g = gv.get(k) * dr[k];
                if (relu && !(gmv[k] * xh + btv[k] > 0.f)) g = 0.f;
                o.set(k, rs[k] * (gmv[k] * g - (pa[k] + xh * pb[k])));
            }
            st16(GX + base + e, o);
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long e = i * VN;
        const Vec16<T> s = ld16(S + base + e), gv = ld16(GY + base + e);
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            const int c = c0 + k;
            const float gm = ss[2 * C + c], bt = ss[3 * C + c], rstd = ss[C + c];
            float xh, g = gv.get(k) * ss[6 * C + c];
            if (from_y) {
                const float y = s.get(k);
                xh = (y - bt) / gm;
                if (relu && !(y > 0.f)) g = 0.f;
            } else {
                xh = (s.get(k) - ss[c]) * rstd;
                if (relu && !(gm * xh + bt > 0.f)) g = 0.f;
            }
            o.set(k, rstd * (gm * g - (ss[4 * C + c] + xh * ss[5 * C + c])));
        }
        st16(GX + base + e, o);
        c0 += cstep;
        if (c0 >= C) c0 -= C;
    }
}

// ------------------------------------------------------------------------------------------------
// Small levels (V <= 2048 voxels, i.e. 12^3 and 6^3): one launch does statistics + apply (forward) or reductions + apply (backward).
// A workgroup owns SPAN = max(cpg, 16 B worth) consecutive channels of one sample (= SPAN/cpg whole groups), walks the
// voxels twice (second pass served by L2) and needs no inter-workgroup reduction.  Replaces 3 launches by 1 at the
// levels where launch latency, not bandwidth, sets the time.
// ------------------------------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ void block_reduce_vec(float (&v)[NV], float* sm /* [4][NV] */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) sm[wv * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = sm[k] + sm[NV + k] + sm[2 * NV + k] + sm[3 * NV + k];
}

// rows per thread of the one-launch norms (template parameter ROWS): 8 x 256 threads = the 2048 rows norm_fused_ok admits; 1 where
// the level has at most 256 voxels (6^3, 7x7x5)

// SLAB = true: x does not exist yet -- it is the ordered sum of the split-K partial slabs of the convolution that feeds this norm
// (conv_k3_tile_kernel / conv_gemm_kernel, fp32 [split][row][channel]) plus the bias.  The statistics pass forms it (the same
// summation order as splitk_finish_kernel), rounds it to T, WRITES it to X (the backward needs the pre-norm tensor) and takes the
// statistics of the rounded values: bit-identical to finish launch + norm launch, with one launch less on the critical chain.
template <typename T, int SPAN, int CPG, bool SLAB, int NF_ROWS>
__global__ __launch_bounds__(256) void norm_fused_fwd_kernel(T* __restrict__ X, T* __restrict__ Y, long long V, int C, int G,
                                                             float eps, float* __restrict__ stats, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int relu, const T* __restrict__ skip,
                                                             const float* __restrict__ chan_scale, float* running_mean,
                                                             float* running_var, float momentum, const float* __restrict__ slab,
                                                             int splits, long long MN, const float* __restrict__ cbias) {
    constexpr int VN = Vec16<T>::N, NVEC = SPAN / VN;
    __shared__ float sm[4 * 2 * SPAN];
    constexpr int cpg = CPG;
    const int n = blockIdx.x, c0 = blockIdx.y * SPAN;      // sample fastest: the spans of one sample share XCDs (see the launch)
    T* xp = X + (long long)n * V * C + c0;
    float acc[2 * SPAN];
#pragma unroll
    for (int k = 0; k < 2 * SPAN; ++k) acc[k] = 0.f;
    // The workgroup's slice (V <= 2048 rows x SPAN channels: norm_fused_ok) stays in REGISTERS between the statistics and the apply
    // pass: every thread requests its NF_ROWS rows up front (fixed trip count, clamped addresses: all loads are in flight together)
    // and the apply pass re-reads nothing.  The loop over v this replaces made ~7 dependent L2 round trips per pass (11 us for 1.7 MB).
    // Rows are summed in the order of that loop (masked rows add +0): same statistics bit for bit.
    Vec16<T> xr[NF_ROWS][NVEC];
#pragma unroll
    for (int i = 0; i < NF_ROWS; ++i) {
        const long long v = threadIdx.x + 256 * i;
        const long long vc = v < V ? v : V - 1;
#pragma unroll
        for (int q = 0; q < NVEC; ++q) {
            if (SLAB) {
                // slabs are span-major (conv_gemm / conv_k3_tile with a deferred finish): [split][8-channel span][row][8] fp32, so the
                // rows of this workgroup's span are one contiguous 32-byte-per-row run in every slab (VN == 8 here: bf16 only)
                const long long Mrows = MN / C;
                const long long off = ((long long)((c0 + q * VN) >> 3) * Mrows + (long long)n * V + vc) * 8;
                float f[VN];
#pragma unroll
                for (int k = 0; k < VN; ++k) f[k] = cbias ? cbias[c0 + q * VN + k] : 0.f;
                for (int z = 0; z < splits; ++z) {
#pragma unroll
                    for (int k = 0; k < VN; k += 4) {
                        const float4 p = *reinterpret_cast<const float4*>(slab + (long long)z * MN + off + k);
                        f[k] += p.x; f[k + 1] += p.y; f[k + 2] += p.z; f[k + 3] += p.w;
                    }
                }
#pragma unroll
                for (int k = 0; k < VN; ++k) xr[i][q].set(k, f[k]);
                if (v < V) st16(xp + v * C + q * VN, xr[i][q]);
            } else {
                xr[i][q] = ld16(xp + vc * C + q * VN);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NF_ROWS; ++i) {
        const bool ok = (long long)threadIdx.x + 256 * i < V;
#pragma unroll
        for (int q = 0; q < NVEC; ++q)
#pragma unroll
            for (int k = 0; k < VN; ++k) { const float f = ok ? xr[i][q].get(k) : 0.f; acc[q * VN + k] += f; acc[SPAN + q * VN + k] += f * f; }
    }
    block_reduce_vec<2 * SPAN>(acc, sm);
    // per-channel scale / shift (every thread computes all SPAN of them: cheap, avoids another LDS round trip)
    float sc[SPAN], sh[SPAN], cs[SPAN];
    const double cnt = (double)V * cpg;
#pragma unroll
    for (int k = 0; k < SPAN; ++k) {
        constexpr int dummy_ = 0; (void)dummy_;
        const int gl = k / cpg;                       // group inside the span (compile-time: registers stay statically indexed)
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int j = 0; j < SPAN; ++j)
            if (j / cpg == gl) { s0 += acc[j]; s1 += acc[SPAN + j]; }
        const double mean = s0 / cnt;
        double var = s1 / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const int c = c0 + k;
        const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        sc[k] = rstd * gm;
        sh[k] = bt - (float)mean * rstd * gm;
        cs[k] = chan_scale ? chan_scale[(long long)n * C + c] : 1.f;
        if (threadIdx.x == 0 && (k % cpg) == 0) {
            const int g = c / cpg;
            stats[((long long)n * G + g) * 2] = (float)mean;
            stats[((long long)n * G + g) * 2 + 1] = rstd;
            if (running_mean && gridDim.x == 1 && G == C) {
                running_mean[g] = (1.f - momentum) * running_mean[g] + momentum * (float)mean;
                const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
                running_var[g] = (1.f - momentum) * running_var[g] + momentum * (float)unb;
            }
        }
    }
    T* yp = Y + (long long)n * V * C + c0;
    const T* sp = skip ? skip + (long long)n * V * C + c0 : nullptr;
    Vec16<T> sr[NF_ROWS][NVEC];
    if (sp) {
#pragma unroll
        for (int i = 0; i < NF_ROWS; ++i) {
            const long long v = threadIdx.x + 256 * i, vc = v < V ? v : V - 1;
#pragma unroll
            for (int q = 0; q < NVEC; ++q) sr[i][q] = ld16(sp + vc * C + q * VN);
        }
    }
#pragma unroll
    for (int i = 0; i < NF_ROWS; ++i) {
        const long long v = threadIdx.x + 256 * i;
        if (v < V) {
#pragma unroll
            for (int q = 0; q < NVEC; ++q) {
                Vec16<T> y;
#pragma unroll
                for (int k = 0; k < VN; ++k) {
                    float f = xr[i][q].get(k) * sc[q * VN + k] + sh[q * VN + k];
                    if (relu) f = f < 0.f ? 0.f : f;
                    f *= cs[q * VN + k];
                    if (sp) f += sr[i][q].get(k);
                    y.set(k, f);
                }
                st16(yp + v * C + q * VN, y);
            }
        }
    }
}

template <typename T, int SPAN, int CPG, int NF_ROWS>
__global__ __launch_bounds__(256) void norm_fused_bwd_kernel(const T* __restrict__ X, const T* __restrict__ GY, T* __restrict__ GX,
                                                             long long V, int C, int G, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int relu, const float* __restrict__ chan_scale,
                                                             float* __restrict__ dpart /* [Nb][C][2] or NULL */) {
    constexpr int VN = Vec16<T>::N, NVEC = SPAN / VN;
    __shared__ float sm[4 * 2 * SPAN];
    constexpr int cpg = CPG;
    const int n = blockIdx.x, c0 = blockIdx.y * SPAN;      // sample fastest: the spans of one sample share XCDs (see the launch)
    const T* xp = X + (long long)n * V * C + c0;
    const T* gp = GY + (long long)n * V * C + c0;
    float mean[SPAN], rstd[SPAN], gm[SPAN], bt[SPAN], cs[SPAN];
#pragma unroll
    for (int k = 0; k < SPAN; ++k) {
        const int c = c0 + k, g = c / cpg;
        mean[k] = stats[((long long)n * G + g) * 2];
        rstd[k] = stats[((long long)n * G + g) * 2 + 1];
        gm[k] = gamma ? gamma[c] : 1.f;
        bt[k] = beta ? beta[c] : 0.f;
        cs[k] = chan_scale ? chan_scale[(long long)n * C + c] : 1.f;
    }
    float acc[2 * SPAN];
#pragma unroll
    for (int k = 0; k < 2 * SPAN; ++k) acc[k] = 0.f;
    // register-resident as norm_fused_fwd_kernel: both operands of every row are requested up front and read ONCE
    Vec16<T> xr[NF_ROWS][NVEC], gr[NF_ROWS][NVEC];
#pragma unroll
    for (int i = 0; i < NF_ROWS; ++i) {
        const long long v = threadIdx.x + 256 * i, vc = v < V ? v : V - 1;
#pragma unroll
        for (int q = 0; q < NVEC; ++q) { xr[i][q] = ld16(xp + vc * C + q * VN); gr[i][q] = ld16(gp + vc * C + q * VN); }
    }
#pragma unroll
    for (int i = 0; i < NF_ROWS; ++i) {
        const bool ok = (long long)threadIdx.x + 256 * i < V;
#pragma unroll
        for (int q = 0; q < NVEC; ++q) {
#pragma unroll
            for (int k = 0; k < VN; ++k) {
                const int j = q * VN + k;
                const float xh = (xr[i][q].get(k) - mean[j]) * rstd[j];
                float g = gr[i][q].get(k) * cs[j];
                if ((relu && !(gm[j] * xh + bt[j] > 0.f)) || !ok) g = 0.f;
                acc[j] += g;
                acc[SPAN + j] += g * xh;              // (g = 0 on masked rows; one fma, as in the loop this replaces: same bits)
            }
        }
    }
    block_reduce_vec<2 * SPAN>(acc, sm);
    if (dpart && threadIdx.x == 0) {     // per-(n, channel) {dbeta, dgamma} contributions (static register indices)
#pragma unroll
        for (int k = 0; k < SPAN; ++k) {
            float* d = dpart + ((long long)n * C + c0 + k) * 2;
            d[0] = acc[k];
            d[1] = acc[SPAN + k];
        }
    }
    float A[SPAN], Bq[SPAN];
    const float inv_cnt = 1.f / ((float)V * (float)cpg);
#pragma unroll
    for (int k = 0; k < SPAN; ++k) {
        const int gl = k / cpg;
        float a = 0.f, bq = 0.f;
#pragma unroll
        for (int j = 0; j < SPAN; ++j)
            if (j / cpg == gl) { a += gm[j] * acc[j]; bq += gm[j] * acc[SPAN + j]; }
        A[k] = a * inv_cnt;
        Bq[k] = bq * inv_cnt;
    }
    T* op = GX + (long long)n * V * C + c0;
#pragma unroll
    for (int i = 0; i < NF_ROWS; ++i) {
        const long long v = threadIdx.x + 256 * i;
        if (v < V) {
#pragma unroll
            for (int q = 0; q < NVEC; ++q) {
                Vec16<T> o;
#pragma unroll
                for (int k = 0; k < VN; ++k) {
                    const int j = q * VN + k;
                    const float xh = (xr[i][q].get(k) - mean[j]) * rstd[j];
                    float g = gr[i][q].get(k) * cs[j];
                    if (relu && !(gm[j] * xh + bt[j] > 0.f)) g = 0.f;
                    o.set(k, rstd[j] * (gm[j] * g - (A[j] + xh * Bq[j])));
                }
                st16(op + v * C + q * VN, o);
            }
        }
    }
}

// dgamma/dbeta[c] = sum over samples of the fused backward's per-(n, c) contributions
__global__ void norm_sum_dparams_kernel(const float* __restrict__ dpart, int Nb, int C, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s0 = 0.f, s1 = 0.f;
    for (int n = 0; n < Nb; ++n) { s0 += dpart[((long long)n * C + c) * 2]; s1 += dpart[((long long)n * C + c) * 2 + 1]; }
    if (dbeta) dbeta[c] = s0;
    if (dgamma) dgamma[c] = s1;
}

static bool norm_fused_ok(int dtype, long long V, int C, int G) {
    const int VN = dtype == DYCON_BF16 ? 8 : 4, cpg = C / G;
    const int span = cpg > VN ? cpg : VN;
    return V <= 2048 && (cpg & (cpg - 1)) == 0 && cpg <= 16 && C % span == 0 && (span == 4 || span == 8 || span == 16);
}

// ------------------------------------------------------------------------------------------------
extern "C" size_t dycon_norm_workspace(int Nb, long long V, int C) {
    const NormPlan p = norm_plan(V);
    return ((size_t)Nb * p.chunks * C * 2 + (size_t)Nb * C * 2 + 64) * sizeof(float);
}

static int norm_check(const char* who, int dtype, int Nb, long long V, int C, int G) {
    const int VN = dtype == DYCON_BF16 ? 8 : 4;
    DYCON_REQUIRE(Nb > 0 && V > 0 && C > 0 && G > 0 && C % G == 0, "%s: bad shape Nb=%d V=%lld C=%d G=%d", who, Nb, V, C, G);
    DYCON_REQUIRE(C % VN == 0 && C / VN <= 256, "%s: C=%d must be a multiple of %d and <= %d", who, C, VN, 256 * VN);
    DYCON_REQUIRE(Nb <= 65535, "%s: Nb too large", who);
    return DYCON_OK;
}

extern "C" int dycon_norm_fwd_is_fused(int dtype, long long V, int C, int G) {
    return (G > 0 && C % G == 0 && norm_fused_ok(dtype, V, C, G)) ? 1 : 0;
}

// One-launch norm of the small levels fed directly by the split-K slabs of the producing convolution (bf16 storage): forms
// x = bias + sum_z slab[z] (written to x_out for the backward), its statistics, and y.  Replaces splitk_finish + dycon_norm_fwd.
extern "C" int dycon_norm_fwd_slab(const float* slab, int splits, const float* conv_bias, void* x_out, void* y, int dtype, int Nb,
                                   long long V, int C, int G, float eps, float* stats, const float* gamma, const float* beta,
                                   int relu, const void* skip, const float* chan_scale, dycon_stream_t stream) {
    DYCON_REQUIRE(slab && x_out && y && stats && splits > 0, "norm_fwd_slab: bad arguments");
    DYCON_REQUIRE(dtype == DYCON_BF16, "norm_fwd_slab: bf16 storage only");
    if (int e = norm_check("norm_fwd_slab", dtype, Nb, V, C, G)) return e;
    DYCON_REQUIRE(norm_fused_ok(dtype, V, C, G), "norm_fwd_slab: this shape is not served by the one-launch norm (ask dycon_norm_fwd_is_fused)");
    const int cpg = C / G, span = cpg > 8 ? cpg : 8;
    const long long MN = (long long)Nb * V * C;
    dim3 grid(Nb, C / span);      // x = sample: workgroups are dealt to the 8 XCDs round-robin by linear id, so the C/span workgroups that
                                    // read the same rows (16 B of every row each) land on Nb-spaced ids -- 2 XCDs at Nb = 4 instead of all 8
                                    // (PMC: 3.6x the algorithmic bytes were fetched, every XCD's L2 pulling the lines for itself)
#define DYCON_NFS_(SP, CP, RW) norm_fused_fwd_kernel<bf16, SP, CP, true, RW><<<grid, 256, 0, stream>>>((bf16*)x_out, (bf16*)y, V, C, G, eps, stats, gamma, beta, relu, (const bf16*)skip, chan_scale, nullptr, nullptr, 0.f, slab, splits, MN, conv_bias)
#define DYCON_NFS(SP, CP) do { if (V <= 256) DYCON_NFS_(SP, CP, 1); else DYCON_NFS_(SP, CP, 8); } while (0)
    if (cpg == 1) DYCON_NFS(8, 1); else if (cpg == 2) DYCON_NFS(8, 2); else if (cpg == 4) DYCON_NFS(8, 4);
    else if (cpg == 8) DYCON_NFS(8, 8); else DYCON_NFS(16, 16);
#undef DYCON_NFS
#undef DYCON_NFS_
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_norm_stats(const void* x, int dtype, int Nb, long long V, int C, int G, float eps, float* stats,
                                float* running_mean, float* running_var, float momentum, float* workspace, size_t ws_bytes,
                                dycon_stream_t stream) {
    DYCON_REQUIRE(x && stats && workspace, "norm_stats: null pointer");
    if (int e = norm_check("norm_stats", dtype, Nb, V, C, G)) return e;
    DYCON_REQUIRE(ws_bytes >= dycon_norm_workspace(Nb, V, C), "norm_stats: workspace too small");
    const NormPlan p = norm_plan(V);
    dim3 grid(p.chunks, Nb);
    DYCON_DISPATCH(dtype, {
        norm_partial_kernel<T, 0><<<grid, 256, 0, stream>>>((const T*)x, nullptr, workspace, V, C, G, p.rows_per_chunk, nullptr,
                                                            nullptr, nullptr, 0, 0, nullptr);
    });
    DYCON_LAUNCH_CHECK();
    norm_finalize_stats_kernel<<<Nb * G, 256, 0, stream>>>(workspace, Nb, p.chunks, C, G, V, eps, stats, running_mean, running_var, momentum);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

// One-launch forward for the small levels; returns DYCON_OK and sets *done = 1 when it handled the call.
extern "C" int dycon_norm_fwd(const void* x, void* y, int dtype, int Nb, long long V, int C, int G, float eps, float* stats,
                              const float* gamma, const float* beta, int relu, const void* skip, const float* chan_scale,
                              float* running_mean, float* running_var, float momentum, float* workspace, size_t ws_bytes,
                              dycon_stream_t stream) {
    DYCON_REQUIRE(x && y && stats, "norm_fwd: null pointer");
    if (int e = norm_check("norm_fwd", dtype, Nb, V, C, G)) return e;
    if (norm_fused_ok(dtype, V, C, G)) {
        const int VN = dtype == DYCON_BF16 ? 8 : 4, cpg = C / G;
        const int span = cpg > VN ? cpg : VN;
        dim3 grid(Nb, C / span);      // (x = sample: see dycon_norm_fwd_slab)
#define DYCON_NF_(TT, SP, CP, RW) norm_fused_fwd_kernel<TT, SP, CP, false, RW><<<grid, 256, 0, stream>>>((TT*)const_cast<void*>(x), (TT*)y, V, C, G, eps, stats, gamma, beta, relu, (const TT*)skip, chan_scale, running_mean, running_var, momentum, nullptr, 0, 0, nullptr)
#define DYCON_NF(TT, SP, CP) do { if (V <= 256) DYCON_NF_(TT, SP, CP, 1); else DYCON_NF_(TT, SP, CP, 8); } while (0)
        if (dtype == DYCON_BF16) {
            if (cpg == 1) DYCON_NF(bf16, 8, 1); else if (cpg == 2) DYCON_NF(bf16, 8, 2); else if (cpg == 4) DYCON_NF(bf16, 8, 4);
            else if (cpg == 8) DYCON_NF(bf16, 8, 8); else DYCON_NF(bf16, 16, 16);
        } else {
            if (cpg == 1) DYCON_NF(float, 4, 1); else if (cpg == 2) DYCON_NF(float, 4, 2); else if (cpg == 4) DYCON_NF(float, 4, 4);
            else if (cpg == 8) DYCON_NF(float, 8, 8); else DYCON_NF(float, 16, 16);
        }
#undef DYCON_NF
#undef DYCON_NF_
        DYCON_LAUNCH_CHECK();
        return DYCON_OK;
    }
    DYCON_REQUIRE(workspace, "norm_fwd: workspace missing");
    if (int e = dycon_norm_stats(x, dtype, Nb, V, C, G, eps, stats, running_mean, running_var, momentum, workspace, ws_bytes, stream)) return e;
    return dycon_norm_apply(x, y, dtype, Nb, V, C, G, stats, gamma, beta, relu, skip, chan_scale, stream);
}

static int apply_grid(long long V, int C, int VN) {
    static const long long vpt = [] { const char* v = getenv("DYCON_NORM_APPLY_VPT"); return v && *v ? atoll(v) : 4LL; }();
    static const long long cap = [] { const char* v = getenv("DYCON_NORM_APPLY_CAP"); return v && *v ? atoll(v) : 2048LL; }();
    long long blocks = (V * C / VN + 256 * vpt - 1) / (256 * vpt);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

extern "C" int dycon_norm_apply(const void* x, void* y, int dtype, int Nb, long long V, int C, int G, const float* stats,
                                const float* gamma, const float* beta, int relu, const void* skip, const float* chan_scale,
                                dycon_stream_t stream) {
    DYCON_REQUIRE(x && y && stats, "norm_apply: null pointer");
    if (int e = norm_check("norm_apply", dtype, Nb, V, C, G)) return e;
    DYCON_DISPATCH(dtype, {
        dim3 grid(apply_grid(V, C, Vec16<T>::N), Nb);
        norm_apply_kernel<T, false><<<grid, 256, 3 * C * sizeof(float), stream>>>((const T*)x, (T*)y, V, C, G, stats, gamma, beta, relu,
                                                                                  (const T*)skip, chan_scale);
    });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

// dycon_norm_fwd on the three-launch shapes WITHOUT its statistics pass: `part` holds per-(n, chunk, channel) {sum, sum of squares}
// partials ([Nb][chunks][C][2] floats) that the producer of x left behind (dycon_conv_gemm_stats); finalize + apply only.
extern "C" int dycon_norm_fwd_parts(const void* x, void* y, int dtype, int Nb, long long V, int C, int G, float eps, float* stats,
                                    const float* gamma, const float* beta, int relu, const void* skip, const float* chan_scale,
                                    float* running_mean, float* running_var, float momentum, const float* part, int chunks,
                                    dycon_stream_t stream) {
    DYCON_REQUIRE(x && y && stats && part && chunks > 0, "norm_fwd_parts: bad arguments");
    if (int e = norm_check("norm_fwd_parts", dtype, Nb, V, C, G)) return e;
    norm_finalize_stats_kernel<<<Nb * G, 256, 0, stream>>>(part, Nb, chunks, C, G, V, eps, stats, running_mean, running_var, momentum);
    DYCON_LAUNCH_CHECK();
    return dycon_norm_apply(x, y, dtype, Nb, V, C, G, stats, gamma, beta, relu, skip, chan_scale, stream);
}

// the finalize launch alone: mean / rstd per (n, g) from a producer's partials (for consumers that apply the statistics themselves:
// dycon_norm_head_fwd, the first block's backward)
extern "C" int dycon_norm_stats_parts(int dtype, int Nb, long long V, int C, int G, float eps, float* stats, float* running_mean,
                                      float* running_var, float momentum, const float* part, int chunks, dycon_stream_t stream) {
    DYCON_REQUIRE(stats && part && chunks > 0, "norm_stats_parts: bad arguments");
    if (int e = norm_check("norm_stats_parts", dtype, Nb, V, C, G)) return e;
    norm_finalize_stats_kernel<<<Nb * G, 256, 0, stream>>>(part, Nb, chunks, C, G, V, eps, stats, running_mean, running_var, momentum);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_norm_bwd(const void* src, int from_y, const void* gy, void* gx, int dtype, int Nb, long long V, int C,
                              int G, const float* stats, const float* gamma, const float* beta, int relu,
                              const float* chan_scale, float* dgamma, float* dbeta, float* workspace, size_t ws_bytes,
                              dycon_stream_t stream) {
    return dycon_norm_bwd_ex(src, from_y, gy, gx, dtype, Nb, V, C, G, stats, gamma, beta, relu, chan_scale, dgamma, dbeta, 0, workspace,
                             ws_bytes, stream);
}

// The first two launches of dycon_norm_bwd alone (three-launch shapes): statistics pass + finalize.  dgamma / dbeta are written and the
// per-group {A, B} sums are left at float offset dycon_norm_bwd_ab_offset(Nb, V, C) of `workspace` for a consumer that forms the data
// gradient itself (dycon_conv1_wgrad_normbwd: the first layer's weight gradient reads gy and z and never sees a stored gz).
extern "C" size_t dycon_norm_bwd_ab_offset(int Nb, long long V, int C) {
    const NormPlan p = norm_plan(V);
    return (size_t)Nb * p.chunks * C * 2;
}
extern "C" int dycon_norm_bwd_stats(const void* src, const void* gy, int dtype, int Nb, long long V, int C, int G, const float* stats,
                                    const float* gamma, const float* beta, int relu, const float* chan_scale, float* dgamma,
                                    float* dbeta, float* workspace, size_t ws_bytes, dycon_stream_t stream) {
    DYCON_REQUIRE(src && gy && stats && workspace, "norm_bwd_stats: null pointer");
    if (int e = norm_check("norm_bwd_stats", dtype, Nb, V, C, G)) return e;
    DYCON_REQUIRE(ws_bytes >= dycon_norm_workspace(Nb, V, C), "norm_bwd_stats: workspace too small");
    const NormPlan p = norm_plan(V);
    float* ab = workspace + dycon_norm_bwd_ab_offset(Nb, V, C);
    dim3 grid(p.chunks, Nb);
    DYCON_DISPATCH(dtype, {
        norm_partial_kernel<T, 1><<<grid, 256, 0, stream>>>((const T*)src, (const T*)gy, workspace, V, C, G, p.rows_per_chunk,
                                                            stats, gamma, beta, relu, 0, chan_scale);
    });
    DYCON_LAUNCH_CHECK();
    const int nfin = Nb * G + ((dgamma || dbeta) ? C : 0);
    norm_finalize_bwd_kernel<<<nfin, 256, 0, stream>>>(workspace, Nb, p.chunks, C, G, gamma, ab, dgamma, dbeta);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

// the per-sample {dbeta, dgamma} contributions a deferred one-launch backward left in its workspace -> dgamma / dbeta
extern "C" int dycon_norm_sum_dparams(const float* workspace, int Nb, int C, float* dgamma, float* dbeta, dycon_stream_t stream) {
    DYCON_REQUIRE(workspace && Nb > 0 && C > 0 && (dgamma || dbeta), "norm_sum_dparams: bad arguments");
    norm_sum_dparams_kernel<<<cdiv(C, 256), 256, 0, stream>>>(workspace, Nb, C, dgamma, dbeta);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_norm_bwd_ex(const void* src, int from_y, const void* gy, void* gx, int dtype, int Nb, long long V, int C,
                                 int G, const float* stats, const float* gamma, const float* beta, int relu,
                                 const float* chan_scale, float* dgamma, float* dbeta, int defer_dparams, float* workspace,
                                 size_t ws_bytes, dycon_stream_t stream) {
    DYCON_REQUIRE(src && gy && gx && stats && workspace, "norm_bwd: null pointer");
    if (int e = norm_check("norm_bwd", dtype, Nb, V, C, G)) return e;
    DYCON_REQUIRE(ws_bytes >= dycon_norm_workspace(Nb, V, C), "norm_bwd: workspace too small");
    if (!from_y && norm_fused_ok(dtype, V, C, G)) {   // small levels: one launch (+ a tiny per-channel sum over the samples)
        const int VN = dtype == DYCON_BF16 ? 8 : 4, cpg = C / G;
        const int span = cpg > VN ? cpg : VN;
        const bool want = dgamma || dbeta;
        dim3 grid(Nb, C / span);      // (x = sample: see dycon_norm_fwd_slab)
#define DYCON_NB_(TT, SP, CP, RW) norm_fused_bwd_kernel<TT, SP, CP, RW><<<grid, 256, 0, stream>>>((const TT*)src, (const TT*)gy, (TT*)gx, V, C, G, stats, gamma, beta, relu, chan_scale, want ? workspace : nullptr)
#define DYCON_NB(TT, SP, CP) do { if (V <= 256) DYCON_NB_(TT, SP, CP, 1); else DYCON_NB_(TT, SP, CP, 8); } while (0)
        if (dtype == DYCON_BF16) {
            if (cpg == 1) DYCON_NB(bf16, 8, 1); else if (cpg == 2) DYCON_NB(bf16, 8, 2); else if (cpg == 4) DYCON_NB(bf16, 8, 4);
            else if (cpg == 8) DYCON_NB(bf16, 8, 8); else DYCON_NB(bf16, 16, 16);
        } else {
            if (cpg == 1) DYCON_NB(float, 4, 1); else if (cpg == 2) DYCON_NB(float, 4, 2); else if (cpg == 4) DYCON_NB(float, 4, 4);
            else if (cpg == 8) DYCON_NB(float, 8, 8); else DYCON_NB(float, 16, 16);
        }
#undef DYCON_NB
#undef DYCON_NB_
        DYCON_LAUNCH_CHECK();
        if (want && !defer_dparams) {     // deferred: the caller sums them (dycon_norm_sum_dparams), possibly on another stream
            norm_sum_dparams_kernel<<<cdiv(C, 256), 256, 0, stream>>>(workspace, Nb, C, dgamma, dbeta);
            DYCON_LAUNCH_CHECK();
        }
        return DYCON_OK;
    }
    const NormPlan p = norm_plan(V);
    float* ab = workspace + (size_t)Nb * p.chunks * C * 2;
    dim3 grid(p.chunks, Nb);
    DYCON_DISPATCH(dtype, {
        norm_partial_kernel<T, 1><<<grid, 256, 0, stream>>>((const T*)src, (const T*)gy, workspace, V, C, G, p.rows_per_chunk,
                                                            stats, gamma, beta, relu, from_y, chan_scale);
    });
    DYCON_LAUNCH_CHECK();
    const int nfin = Nb * G + ((dgamma || dbeta) ? C : 0);
    norm_finalize_bwd_kernel<<<nfin, 256, 0, stream>>>(workspace, Nb, p.chunks, C, G, gamma, ab, dgamma, dbeta);
    DYCON_LAUNCH_CHECK();
    DYCON_DISPATCH(dtype, {
        dim3 grid2(apply_grid(V, C, Vec16<T>::N), Nb);
        norm_bwd_apply_kernel<T, false><<<grid2, 256, 7 * C * sizeof(float), stream>>>((const T*)src, (const T*)gy, (T*)gx, V, C, G, stats,
                                                                                       gamma, beta, ab, relu, from_y, chan_scale);
    });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

// ------------------------------------------------------------------------------------------------
// The 2-class head fused into the LAST normalisation of the V-Net (block_nine: conv -> norm -> ReLU [-> Dropout3d] -> out_conv 1x1,
// VNet.py:225-227).  At 96^3 the normalised 16-channel tensor y is the largest activation of the step and its only consumers are a
// 16 -> 2 matrix product forward, the 2 -> 16 product that makes its gradient, and the 2 x 16 weight gradient.  So y and its
// gradient are never materialised:
//   forward : norm statistics (dycon_norm_stats), then ONE pass  x -> y (registers) -> logits          (was: apply + 1x1 head)
//   backward: the two passes of the norm backward form  gy = g_logits . W  per voxel on the fly and the statistics pass also
//             accumulates dW = sum g_logits (x) y and db = sum g_logits                      (was: + head dgrad + head wgrad)
// Arithmetic per voxel is that of norm_apply_kernel / head_1x1_fwd_kernel / head_1x1_bwd_kernel, including the rounding of y and
// gy to the storage type: the logits are the unfused path's bit for bit, gz to fp32 round-off; dW / db differ in summation order.
// ------------------------------------------------------------------------------------------------
constexpr int HC = 16, HK = 2;                      // channels in, logits out
constexpr int HPART = HK * HC + HK;                 // per-chunk head partials: dW[k][c], db[k]

template <typename T> __device__ __forceinline__ float round_to(float v) { T t; stf(&t, v); return ldf(&t); }

template <typename T>
__global__ __launch_bounds__(256) void norm_apply_head_kernel(const T* __restrict__ X, long long V, int G,
                                                              const float* __restrict__ stats, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int relu,
                                                              const float* __restrict__ chan_scale, const float* __restrict__ W,
                                                              const float* __restrict__ hb, float* __restrict__ L) {
    constexpr int VN = Vec16<T>::N;
    __shared__ float ss[3 * HC + HK * HC + HK];
    const int n = blockIdx.y, cpg = HC / G;
    if (threadIdx.x < HC) {
        const int c = threadIdx.x, g = c / cpg;
        const float mean = stats[((long long)n * G + g) * 2], rstd = stats[((long long)n * G + g) * 2 + 1];
        const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        ss[c] = rstd * gm;
        ss[HC + c] = bt - mean * rstd * gm;
        ss[2 * HC + c] = chan_scale ? chan_scale[(long long)n * HC + c] : 1.f;
    }
    if (threadIdx.x < HK * HC) ss[3 * HC + threadIdx.x] = W[threadIdx.x];
    if (threadIdx.x < HK) ss[3 * HC + HK * HC + threadIdx.x] = hb ? hb[threadIdx.x] : 0.f;
    __syncthreads();
    float sc[HC], sh[HC], dr[HC], w0[HC], w1[HC];
#pragma unroll
    for (int c = 0; c < HC; ++c) { sc[c] = ss[c]; sh[c] = ss[HC + c]; dr[c] = ss[2 * HC + c]; w0[c] = ss[3 * HC + c]; w1[c] = ss[4 * HC + c]; }
    const float b0 = ss[3 * HC + HK * HC], b1 = ss[3 * HC + HK * HC + 1];
    const T* xp = X + (long long)n * V * HC;
    float2* lp = reinterpret_cast<float2*>(L) + (long long)n * V;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long long)gridDim.x * 256) {
        float o0 = b0, o1 = b1;
#pragma unroll
        for (int c0 = 0; c0 < HC; c0 += VN) {
            const Vec16<T> x = ld16(xp + v * HC + c0);
#pragma unroll
            for (int k = 0; k < VN; ++k) {
                const int c = c0 + k;
                float y = x.get(k) * sc[c] + sh[c];
                if (relu) y = y < 0.f ? 0.f : y;
                y = round_to<T>(y * dr[c]);          // what norm_apply_kernel would have stored
                o0 += y * w0[c];
                o1 += y * w1[c];
            }
        }
        lp[v] = make_float2(o0, o1);
    }
}

// statistics pass of the backward: thread mapping, row order and summation order of norm_partial_kernel<T, 1>, with gy formed on the fly;
// additionally the head's weight / bias gradient partials of the chunk (hpart[(n, chunk)][HPART])
template <typename T>
__global__ __launch_bounds__(256) void norm_head_partial_kernel(const T* __restrict__ S, const float* __restrict__ GL,
                                                                float* __restrict__ part, float* __restrict__ hpart, long long V,
                                                                int G, long long rows_per_chunk, const float* __restrict__ stats,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                int relu, const float* __restrict__ chan_scale,
                                                                const float* __restrict__ W) {
    constexpr int VN = Vec16<T>::N;
    constexpr int NGRP = HC / VN, RPI = 256 / NGRP;
    __shared__ float sm[4][256][VN + 1];
    __shared__ float sb[256][HK];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int cg = threadIdx.x % NGRP, rr = threadIdx.x / NGRP;
    const long long v0 = chunk * rows_per_chunk, v1 = min(V, v0 + rows_per_chunk);
    const int cpg = HC / G;
    float a0[VN], a1[VN], d0[VN], d1[VN], e0 = 0.f, e1 = 0.f;
    float mean[VN], rstd[VN], gm[VN], bt[VN], cs[VN], sc[VN], sh[VN], w0[VN], w1[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) {
        const int c = cg * VN + k, g = c / cpg;
        a0[k] = a1[k] = d0[k] = d1[k] = 0.f;
        cs[k] = chan_scale ? chan_scale[(long long)n * HC + c] : 1.f;
        mean[k] = stats[((long long)n * G + g) * 2];
        rstd[k] = stats[((long long)n * G + g) * 2 + 1];
        gm[k] = gamma ? gamma[c] : 1.f;
        bt[k] = beta ? beta[c] : 0.f;
        sc[k] = rstd[k] * gm[k];
        sh[k] = bt[k] - mean[k] * rstd[k] * gm[k];
        w0[k] = W[c];
        w1[k] = W[HC + c];
    }
    const T* sp = S + ((long long)n * V) * HC + cg * VN;
    const float2* gp = reinterpret_cast<const float2*>(GL) + (long long)n * V;
    for (long long v = v0 + rr; v < v1; v += RPI) {
        const Vec16<T> s = ld16(sp + v * HC);
        const float2 gl = gp[v];
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            float gyv = 0.f;                         // head_1x1_bwd_kernel's order and rounding
            gyv += gl.x * w0[k];
            gyv += gl.y * w1[k];
            float g = round_to<T>(gyv) * cs[k];
            const float x = s.get(k);
            const float xh = (x - mean[k]) * rstd[k];
            if (relu && !(gm[k] * xh + bt[k] > 0.f)) g = 0.f;
            a0[k] += g;
            a1[k] += g * xh;
            float y = x * sc[k] + sh[k];             // the forward's y (norm_apply_head_kernel)
            if (relu) y = y < 0.f ? 0.f : y;
            y = round_to<T>(y * cs[k]);
            d0[k] += gl.x * y;
            d1[k] += gl.y * y;
        }
        if (cg == 0) { e0 += gl.x; e1 += gl.y; }
    }
#pragma unroll
    for (int k = 0; k < VN; ++k) {
        sm[0][threadIdx.x][k] = a0[k]; sm[1][threadIdx.x][k] = a1[k];
        sm[2][threadIdx.x][k] = d0[k]; sm[3][threadIdx.x][k] = d1[k];
    }
    sb[threadIdx.x][0] = e0; sb[threadIdx.x][1] = e1;
    __syncthreads();
    if (threadIdx.x < HC) {                          // {sum g, sum g*xhat} per channel, as norm_partial_kernel stores them
        const int o = threadIdx.x, og = o / VN, ok = o % VN;
        float s0 = 0.f, s1 = 0.f;
        for (int q = 0; q < RPI; ++q) { s0 += sm[0][q * NGRP + og][ok]; s1 += sm[1][q * NGRP + og][ok]; }
        float* d = part + ((((long long)n * gridDim.x + chunk) * HC) + o) * 2;
        d[0] = s0;
        d[1] = s1;
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + HK * HC) {
        const int j = threadIdx.x - 64, kk = j / HC, o = j % HC, og = o / VN, ok = o % VN;
        float s0 = 0.f;
        for (int q = 0; q < RPI; ++q) s0 += sm[2 + kk][q * NGRP + og][ok];
        hpart[((long long)n * gridDim.x + chunk) * HPART + j] = s0;
    } else if (threadIdx.x >= 128 && threadIdx.x < 128 + HK) {
        const int kk = threadIdx.x - 128;
        float s0 = 0.f;
        for (int q = 0; q < 256; ++q) s0 += sb[q][kk];
        hpart[((long long)n * gridDim.x + chunk) * HPART + HK * HC + kk] = s0;
    }
}

// sums of the head partials over all (n, chunk): one workgroup per output (HPART of them)
__global__ __launch_bounds__(256) void norm_head_finalize_kernel(const float* __restrict__ hpart, int items, float* __restrict__ dW,
                                                                 float* __restrict__ db) {
    const int j = blockIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int e = threadIdx.x; e < items; e += 256) s0 += hpart[(long long)e * HPART + j];
    block_sum2_double(s0, s1);
    if (threadIdx.x == 0) {
        if (j < HK * HC) { if (dW) dW[j] = (float)s0; }
        else if (db) db[j - HK * HC] = (float)s0;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void norm_head_bwd_apply_kernel(const T* __restrict__ S, const float* __restrict__ GL,
                                                                  T* __restrict__ GX, long long V, int G,
                                                                  const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, const float* __restrict__ ab,
                                                                  int relu, const float* __restrict__ chan_scale,
                                                                  const float* __restrict__ W) {
    constexpr int VN = Vec16<T>::N;
    constexpr int NGRP = HC / VN;
    const int n = blockIdx.y, cpg = HC / G;
    const float inv_cnt = 1.f / ((float)V * (float)cpg);
    const long long total = V * NGRP;
    // 256 * VN is a multiple of 16: a thread's channels never change
    const int cg = threadIdx.x % NGRP;
    float mu[VN], rs[VN], gmv[VN], btv[VN], pa[VN], pb[VN], dr[VN], w0[VN], w1[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) {
        const int c = cg * VN + k, g = c / cpg;
        mu[k] = stats[((long long)n * G + g) * 2];
        rs[k] = stats[((long long)n * G + g) * 2 + 1];
        gmv[k] = gamma ? gamma[c] : 1.f;
        btv[k] = beta ? beta[c] : 0.f;
        pa[k] = ab[((long long)n * G + g) * 2] * inv_cnt;
        pb[k] = ab[((long long)n * G + g) * 2 + 1] * inv_cnt;
        dr[k] = chan_scale ? chan_scale[(long long)n * HC + c] : 1.f;
        w0[k] = W[c];
        w1[k] = W[HC + c];
    }
    const long long base = (long long)n * V * HC;
    const float2* gp = reinterpret_cast<const float2*>(GL) + (long long)n * V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const Vec16<T> s = ld16(S + base + i * VN);
        const float2 gl = gp[i / NGRP];
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            float gyv = 0.f;
            gyv += gl.x * w0[k];
            gyv += gl.y * w1[k];
            float g = round_to<T>(gyv) * dr[k];
            const float xh = (s.get(k) - mu[k]) * rs[k];
            if (relu && !(gmv[k] * xh + btv[k] > 0.f)) g = 0.f;
            o.set(k, rs[k] * (gmv[k] * g - (pa[k] + xh * pb[k])));
        }
        st16(GX + base + i * VN, o);
    }
}

extern "C" size_t dycon_norm_head_workspace(int Nb, long long V) {
    const NormPlan p = norm_plan(V);
    return dycon_norm_workspace(Nb, V, HC) + (size_t)Nb * p.chunks * HPART * sizeof(float);
}

extern "C" int dycon_norm_head_fwd(const void* x, int dtype, int Nb, long long V, int G, const float* stats, const float* gamma,
                                   const float* beta, int relu, const float* chan_scale, const float* head_w, const float* head_b,
                                   float* logits, dycon_stream_t stream) {
    DYCON_REQUIRE(x && stats && head_w && logits, "norm_head_fwd: null pointer");
    if (int e = norm_check("norm_head_fwd", dtype, Nb, V, HC, G)) return e;
    long long blocks = (V + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    dim3 grid((int)blocks, Nb);
    DYCON_DISPATCH(dtype, {
        norm_apply_head_kernel<T><<<grid, 256, 0, stream>>>((const T*)x, V, G, stats, gamma, beta, relu, chan_scale, head_w, head_b, logits);
    });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_norm_head_bwd(const void* x, const float* g_logits, void* gx, int dtype, int Nb, long long V, int G,
                                   const float* stats, const float* gamma, const float* beta, int relu, const float* chan_scale,
                                   const float* head_w, float* dgamma, float* dbeta, float* workspace, size_t ws_bytes,
                                   dycon_stream_t stream) {
    DYCON_REQUIRE(x && g_logits && gx && stats && head_w && workspace, "norm_head_bwd: null pointer");
    if (int e = norm_check("norm_head_bwd", dtype, Nb, V, HC, G)) return e;
    DYCON_REQUIRE(ws_bytes >= dycon_norm_head_workspace(Nb, V), "norm_head_bwd: workspace too small");
    const NormPlan p = norm_plan(V);
    float* ab = workspace + (size_t)Nb * p.chunks * HC * 2;
    float* hpart = workspace + dycon_norm_workspace(Nb, V, HC) / sizeof(float);
    dim3 grid(p.chunks, Nb);
    DYCON_DISPATCH(dtype, {
        norm_head_partial_kernel<T><<<grid, 256, 0, stream>>>((const T*)x, g_logits, workspace, hpart, V, G, p.rows_per_chunk, stats, gamma,
                                                              beta, relu, chan_scale, head_w);
    });
    DYCON_LAUNCH_CHECK();
    const int nfin = Nb * G + ((dgamma || dbeta) ? HC : 0);
    norm_finalize_bwd_kernel<<<nfin, 256, 0, stream>>>(workspace, Nb, p.chunks, HC, G, gamma, ab, dgamma, dbeta);
    DYCON_LAUNCH_CHECK();
    DYCON_DISPATCH(dtype, {
        dim3 grid2(apply_grid(V, HC, Vec16<T>::N), Nb);
        norm_head_bwd_apply_kernel<T><<<grid2, 256, 0, stream>>>((const T*)x, g_logits, (T*)gx, V, G, stats, gamma, beta, ab, relu, chan_scale,
                                                                 head_w);
    });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

// the head's weight / bias gradient from the partials dycon_norm_head_bwd left in its workspace (a tiny launch the caller may put on
// another stream: nothing on the data-gradient chain depends on it)
extern "C" int dycon_norm_head_dparams(const float* workspace, int Nb, long long V, float* d_head_w, float* d_head_b,
                                       dycon_stream_t stream) {
    DYCON_REQUIRE(workspace && (d_head_w || d_head_b), "norm_head_dparams: bad arguments");
    const NormPlan p = norm_plan(V);
    const float* hpart = workspace + dycon_norm_workspace(Nb, V, HC) / sizeof(float);
    norm_head_finalize_kernel<<<HPART, 256, 0, stream>>>(hpart, Nb * p.chunks, d_head_w, d_head_b);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

// ------------------------------------------------------------------------------------------------
// Accumulator forms: the caller provides `acc`, dycon_norm_acc_doubles() doubles that are ZERO on entry (a slice of an arena it
// clears once per step; Nb x nslots x C x 2: the chunks of a sample are spread over nslots copies of its accumulators).
// Shapes served by the one-launch kernels (dycon_norm_fwd_is_fused) ignore it; the others run statistics + apply as TWO launches
// (no finalize): see norm_partial_kernel.  Results equal dycon_norm_fwd / dycon_norm_bwd up to the last bits of the double sums.
// ------------------------------------------------------------------------------------------------
static int norm_slots(long long V) {
    const NormPlan p = norm_plan(V);
    return p.chunks < 32 ? p.chunks : 32;
}
extern "C" size_t dycon_norm_acc_doubles(int Nb, long long V, int C) { return (size_t)Nb * norm_slots(V) * C * 2; }

extern "C" int dycon_norm_fwd_acc(const void* x, void* y, int dtype, int Nb, long long V, int C, int G, float eps, float* stats,
                                  const float* gamma, const float* beta, int relu, const void* skip, const float* chan_scale,
                                  float* running_mean, float* running_var, float momentum, double* acc, dycon_stream_t stream) {
    DYCON_REQUIRE(x && y && stats, "norm_fwd_acc: null pointer");
    if (int e = norm_check("norm_fwd_acc", dtype, Nb, V, C, G)) return e;
    if (norm_fused_ok(dtype, V, C, G))
        return dycon_norm_fwd(x, y, dtype, Nb, V, C, G, eps, stats, gamma, beta, relu, skip, chan_scale, running_mean, running_var,
                              momentum, nullptr, 0, stream);
    DYCON_REQUIRE(acc, "norm_fwd_acc: accumulator missing");
    const NormPlan p = norm_plan(V);
    dim3 grid(p.chunks, Nb);
    DYCON_DISPATCH(dtype, {
        norm_partial_kernel<T, 0><<<grid, 256, 0, stream>>>((const T*)x, nullptr, nullptr, V, C, G, p.rows_per_chunk, nullptr, nullptr,
                                                            nullptr, 0, 0, nullptr, acc, norm_slots(V));
        dim3 grid2(apply_grid(V, C, Vec16<T>::N), Nb);
        norm_apply_kernel<T, true><<<grid2, 256, 3 * C * sizeof(float), stream>>>((const T*)x, (T*)y, V, C, G, nullptr, gamma, beta, relu,
                                                                                 (const T*)skip, chan_scale, acc, stats, eps,
                                                                                 running_mean, running_var, momentum, norm_slots(V));
    });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_norm_bwd_acc(const void* src, const void* gy, void* gx, int dtype, int Nb, long long V, int C, int G,
                                  const float* stats, const float* gamma, const float* beta, int relu, const float* chan_scale,
                                  float* dgamma, float* dbeta, double* acc, float* workspace, size_t ws_bytes,
                                  dycon_stream_t stream) {
    DYCON_REQUIRE(src && gy && gx && stats, "norm_bwd_acc: null pointer");
    if (int e = norm_check("norm_bwd_acc", dtype, Nb, V, C, G)) return e;
    if (norm_fused_ok(dtype, V, C, G))
        return dycon_norm_bwd(src, 0, gy, gx, dtype, Nb, V, C, G, stats, gamma, beta, relu, chan_scale, dgamma, dbeta, workspace,
                              ws_bytes, stream);
    DYCON_REQUIRE(acc, "norm_bwd_acc: accumulator missing");
    const NormPlan p = norm_plan(V);
    dim3 grid(p.chunks, Nb);
    DYCON_DISPATCH(dtype, {
        norm_partial_kernel<T, 1><<<grid, 256, 0, stream>>>((const T*)src, (const T*)gy, nullptr, V, C, G, p.rows_per_chunk, stats, gamma,
                                                            beta, relu, 0, chan_scale, acc, norm_slots(V));
        dim3 grid2(apply_grid(V, C, Vec16<T>::N), Nb);
        norm_bwd_apply_kernel<T, true><<<grid2, 256, 7 * C * sizeof(float), stream>>>((const T*)src, (const T*)gy, (T*)gx, V, C, G, stats,
                                                                                     gamma, beta, nullptr, relu, 0, chan_scale, acc,
                                                                                     dgamma, dbeta, norm_slots(V));
    });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}
