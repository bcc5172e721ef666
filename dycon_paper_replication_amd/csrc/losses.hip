// Loss kernels of the DyCON step: the fused voxel losses (CE, Dice, consistency, UnCL), the
// embedding row-normalisation, the contrastive mask, and the blockwise FeCL.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// =================================================================================================
// voxel losses, 2 classes.  logits are (B, V, 2) fp32 (NDHWC with C = 2)
// =================================================================================================
__device__ __forceinline__ int load_label(const void* labels, int label_bytes, long long i) {
    return label_bytes == 8 ? (int)((const long long*)labels)[i] : (int)((const uint8_t*)labels)[i];
}

// FAST: the bare hardware v_exp_f32 / v_log_f32 / v_rcp_f32 (1 ulp) instead of the libm routines; the bf16 step uses them (its logits
// carry bf16 noise), the fp32 parity mode does not.  The voxel-loss kernels and the FeCL pair epilogue are bound by these sequences,
// not by their traffic.  (Until round 3 FAST meant __expf / __logf / __fdividef / __frcp_rn, which the compiler expands with
// denormal-range scaling -- v_ldexp, v_cmp_class, v_cndmask around every v_exp / v_log -- and, for the divisions, the full IEEE
// sequence: ten instructions where one v_rcp_f32 does.)  Arguments outside the normal range only occur where the result is immaterial:
// exp of a logit difference below -87 flushes to 0 instead of a denormal; every log / rcp argument here is >= 1e-18.
template <bool FAST> __device__ __forceinline__ float fexp(float x) { return FAST ? __builtin_amdgcn_exp2f(x * 1.4426950408889634f) : expf(x); }
template <bool FAST> __device__ __forceinline__ float flog(float x) { return FAST ? __builtin_amdgcn_logf(x) * 0.6931471805599453f : logf(x); }
template <bool FAST> __device__ __forceinline__ float fdiv(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }

struct Soft2 { float p0, p1, lse, m; };
template <bool FAST> __device__ __forceinline__ Soft2 softmax2(float l0, float l1) {
    Soft2 s;
    s.m = fmaxf(l0, l1);
    const float e0 = fexp<FAST>(l0 - s.m), e1 = fexp<FAST>(l1 - s.m), z = e0 + e1;
    s.p0 = fdiv<FAST>(e0, z);
    s.p1 = fdiv<FAST>(e1, z);
    s.lse = flog<FAST>(z);
    return s;
}

constexpr int NSUM = 11;

template <bool FAST>
__global__ __launch_bounds__(256) void seg_losses_fwd_kernel(const float2* __restrict__ SL, const float2* __restrict__ TL,
                                                             const void* __restrict__ labels, int label_bytes, int B, int LB,
                                                             long long V, float beta, double* __restrict__ sums) {
    __shared__ float red[4][NSUM];
    float acc[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) acc[k] = 0.f;
    const long long total = (long long)B * V, lab_end = (long long)LB * V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float2 ls = SL[i], lt = TL[i];
        const Soft2 s = softmax2<FAST>(ls.x, ls.y), t = softmax2<FAST>(lt.x, lt.y);
        if (i < lab_end) {
            const int y = load_label(labels, label_bytes, i);
            const float ly = y == 1 ? ls.y : ls.x;
            acc[0] += -(ly - s.m - s.lse);
            const float t1 = y == 1 ? 1.f : 0.f, t0 = y == 0 ? 1.f : 0.f;
            acc[1] += s.p1 * t1; acc[2] += s.p1 * s.p1; acc[3] += t1;
            acc[4] += s.p0 * t0; acc[5] += s.p0 * s.p0; acc[6] += t0;
        } else {
            // the reference feeds PROBABILITIES to softmax_mse_loss / softmax_kl_loss, which softmax again
            const Soft2 qs = softmax2<FAST>(s.p0, s.p1), qt = softmax2<FAST>(t.p0, t.p1);
            const float d0 = qs.p0 - qt.p0, d1 = qs.p1 - qt.p1;
            acc[7] += d0 * d0 + d1 * d1;
            const float lqs0 = s.p0 - qs.m - qs.lse, lqs1 = s.p1 - qs.m - qs.lse;
            acc[8] += qt.p0 * (flog<FAST>(qt.p0) - lqs0) + qt.p1 * (flog<FAST>(qt.p1) - lqs1);
        }
        const float hs = -(s.p0 * flog<FAST>(s.p0 + 1e-6f) + s.p1 * flog<FAST>(s.p1 + 1e-6f));
        const float ht = -(t.p0 * flog<FAST>(t.p0 + 1e-6f) + t.p1 * flog<FAST>(t.p1 + 1e-6f));
        const float w = fexp<FAST>(beta * hs) + fexp<FAST>(beta * ht);
        const float e0 = s.p0 - t.p0, e1 = s.p1 - t.p1;
        acc[9] += fdiv<FAST>(e0 * e0 + e1 * e1, w);
        acc[10] += hs + ht;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NSUM; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) red[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NSUM) {
        const double v = (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x];
        atomicAdd(&sums[threadIdx.x], v);
    }
}

template <bool FAST>
__global__ __launch_bounds__(256) void seg_losses_bwd_kernel(const float2* __restrict__ SL, const float2* __restrict__ TL,
                                                             const void* __restrict__ labels, int label_bytes, int B, int LB,
                                                             long long V, float beta, const double* __restrict__ sums,
                                                             const float* __restrict__ coef, int cons_kind,
                                                             float2* __restrict__ G) {
    const float c_ce = coef[0], c_dfg = coef[1], c_dmc = coef[2], c_cons = coef[3], c_uncl = coef[4];
    const float smooth = 1e-5f;
    const float I1 = (float)sums[1], D1 = (float)sums[2] + (float)sums[3] + smooth;
    const float I0 = (float)sums[4], D0 = (float)sums[5] + (float)sums[6] + smooth;
    const float inv_ce = LB > 0 ? 1.f / ((float)LB * (float)V) : 0.f;
    const float inv_cons = B > LB ? 1.f / ((float)(B - LB) * (float)V * 2.f) : 0.f;
    const float inv_all = 1.f / ((float)B * (float)V);
    const float inv_D1sq = 1.f / (D1 * D1), inv_D0sq = 1.f / (D0 * D0);
    const long long total = (long long)B * V, lab_end = (long long)LB * V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float2 ls = SL[i], lt = TL[i];
        const Soft2 s = softmax2<FAST>(ls.x, ls.y), t = softmax2<FAST>(lt.x, lt.y);
        float gp0 = 0.f, gp1 = 0.f;   // d loss / d student probabilities
        float gl0 = 0.f, gl1 = 0.f;   // direct d loss / d logits (cross entropy)
        if (i < lab_end) {
            const int y = load_label(labels, label_bytes, i);
            const float t1 = y == 1 ? 1.f : 0.f, t0 = y == 0 ? 1.f : 0.f;
            gl0 = c_ce * inv_ce * (s.p0 - t0);
            gl1 = c_ce * inv_ce * (s.p1 - t1);
            const float dd1 = -(2.f * t1 * D1 - (2.f * I1 + smooth) * 2.f * s.p1) * inv_D1sq;
            const float dd0 = -(2.f * t0 * D0 - (2.f * I0 + smooth) * 2.f * s.p0) * inv_D0sq;
            gp1 += c_dfg * dd1 + 0.5f * c_dmc * dd1;
            gp0 += 0.5f * c_dmc * dd0;
        } else {
            const Soft2 qs = softmax2<FAST>(s.p0, s.p1), qt = softmax2<FAST>(t.p0, t.p1);
            float gq0, gq1;   // gradient w.r.t. the inner softmax INPUT (= student probabilities)
            if (cons_kind == 0) {
                const float a0 = 2.f * (qs.p0 - qt.p0) * inv_cons, a1 = 2.f * (qs.p1 - qt.p1) * inv_cons;
                const float dot = qs.p0 * a0 + qs.p1 * a1;
                gq0 = qs.p0 * (a0 - dot);
                gq1 = qs.p1 * (a1 - dot);
            } else {
                gq0 = (qs.p0 - qt.p0) * inv_cons;
                gq1 = (qs.p1 - qt.p1) * inv_cons;
            }
            gp0 += c_cons * gq0;
            gp1 += c_cons * gq1;
        }
        {
            const float eps = 1e-6f;
            const float hs = -(s.p0 * flog<FAST>(s.p0 + eps) + s.p1 * flog<FAST>(s.p1 + eps));
            const float ht = -(t.p0 * flog<FAST>(t.p0 + eps) + t.p1 * flog<FAST>(t.p1 + eps));
            const float ehs = fexp<FAST>(beta * hs), w = ehs + fexp<FAST>(beta * ht);
            const float e0 = s.p0 - t.p0, e1 = s.p1 - t.p1, d2 = e0 * e0 + e1 * e1;
            const float dh0 = -(flog<FAST>(s.p0 + eps) + fdiv<FAST>(s.p0, s.p0 + eps)), dh1 = -(flog<FAST>(s.p1 + eps) + fdiv<FAST>(s.p1, s.p1 + eps));
            const float rw = fdiv<FAST>(1.f, w);
            const float k = beta - d2 * beta * ehs * rw * rw;
            gp0 += c_uncl * inv_all * (2.f * e0 * rw + dh0 * k);
            gp1 += c_uncl * inv_all * (2.f * e1 * rw + dh1 * k);
        }
        const float dot = s.p0 * gp0 + s.p1 * gp1;
        G[i] = make_float2(gl0 + s.p0 * (gp0 - dot), gl1 + s.p1 * (gp1 - dot));
    }
}

// scalars of the voxel losses from the accumulated sums (device-side: no host round trip)
//   vals: 0 ce | 1 dice (class 1, losses.dice_loss) | 2 dice (mean over classes, losses.DiceLoss) | 3 cons mse | 4 cons kl | 5 uncl
__global__ void seg_losses_finalize_kernel(const double* __restrict__ sums, int B, int LB, long long V, float beta,
                                           float* __restrict__ vals) {
    const double s = 1e-5;
    const double nl = (double)LB * (double)V, nc = (double)(B - LB) * (double)V * 2.0, na = (double)B * (double)V;
    const double d1 = 1.0 - (2.0 * sums[1] + s) / (sums[2] + sums[3] + s);
    const double d0 = 1.0 - (2.0 * sums[4] + s) / (sums[5] + sums[6] + s);
    vals[0] = LB > 0 ? (float)(sums[0] / nl) : 0.f;
    vals[1] = (float)d1;
    vals[2] = (float)(0.5 * (d0 + d1));
    vals[3] = B > LB ? (float)(sums[7] / nc) : 0.f;
    vals[4] = B > LB ? (float)(sums[8] / nc) : 0.f;
    vals[5] = (float)(sums[9] / na + (double)beta * sums[10] / na);
}

// total = l_w*(ce + dice) + cons_w*cons + u_w*(fecl + uncl)   (train_DyCON_BraTS19.py:355-357) and the NaN/Inf flag (:360-362)
__global__ void step_loss_kernel(const float* __restrict__ vals, const float* __restrict__ fecl, float l_w, float cons_w,
                                 float u_w, int dice_kind, int cons_kind, float* __restrict__ out, int* __restrict__ nonfinite) {
    const float ce = vals[0], dice = vals[dice_kind ? 2 : 1], cons = vals[cons_kind ? 4 : 3], un = vals[5];
    const float fe = fecl ? fecl[0] : 0.f;
    const float total = l_w * (ce + dice) + cons_w * cons + u_w * (fe + un);
    out[0] = total; out[1] = ce; out[2] = dice; out[3] = cons; out[4] = fe; out[5] = un;
    if (nonfinite) nonfinite[0] = isfinite(total) ? 0 : 1;
}

// seg_losses_finalize + fecl_finalize + step_loss in one launch (same arithmetic, same order)
__global__ void step_losses_kernel(const double* __restrict__ sums, const double* __restrict__ fo, int B, int LB, long long V,
                                   float beta, double fecl_rows, float lambda_cross, int has_teacher, float l_w, float cons_w,
                                   float u_w, int dice_kind, int cons_kind, float* __restrict__ out, int* __restrict__ nonfinite) {
    const double s = 1e-5;
    const double nl = (double)LB * (double)V, nc = (double)(B - LB) * (double)V * 2.0, na = (double)B * (double)V;
    const double d1 = 1.0 - (2.0 * sums[1] + s) / (sums[2] + sums[3] + s);
    const double d0 = 1.0 - (2.0 * sums[4] + s) / (sums[5] + sums[6] + s);
    const float ce = LB > 0 ? (float)(sums[0] / nl) : 0.f;
    const float dice = dice_kind ? (float)(0.5 * (d0 + d1)) : (float)d1;
    const float cons = B > LB ? (float)(sums[cons_kind ? 8 : 7] / nc) : 0.f;
    const float un = (float)(sums[9] / na + (double)beta * sums[10] / na);
    float fe = 0.f;
    if (fo) {
        double l = fo[0] / fecl_rows;
        if (has_teacher) l += (double)lambda_cross * fo[1] / (fo[2] + 1e-18);
        fe = (float)l;
    }
    const float total = l_w * (ce + dice) + cons_w * cons + u_w * (fe + un);
    out[0] = total; out[1] = ce; out[2] = dice; out[3] = cons; out[4] = fe; out[5] = un;
    if (nonfinite) nonfinite[0] = isfinite(total) ? 0 : 1;
}

// =================================================================================================
// row L2 normalisation: one wave per row
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ X, T* __restrict__ Y, float* __restrict__ norms,
                                                         long long R, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = ldf(X + row * C + c); ss += v * v; }
    ss = wave_sum(ss);
    const float nrm = fmaxf(sqrtf(ss), eps);
    if (lane == 0) norms[row] = nrm;
    for (int c = lane; c < C; c += 64) stf(Y + row * C + c, ldf(X + row * C + c) / nrm);
}

template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ Y, const float* __restrict__ norms,
                                                         const T* __restrict__ GY, T* __restrict__ GX, long long R, int C,
                                                         float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) dot += ldf(Y + row * C + c) * ldf(GY + row * C + c);
    dot = wave_sum(dot);
    const float nrm = norms[row];
    // below the clamp (||x|| < eps) the denominator is the constant eps: no projection term
    const float proj = nrm > eps ? dot : 0.f;
    for (int c = lane; c < C; c += 64) stf(GX + row * C + c, (ldf(GY + row * C + c) - ldf(Y + row * C + c) * proj) / nrm);
}

// one wave per pooled cell: the 64 lanes walk the cell's kd*kh*kw voxels (x fastest: coalesced) and add up with shuffles --
// a thread per cell would read its 512 voxels serially with a 512-byte stride between neighbouring threads
__global__ __launch_bounds__(256) void mask_pool_kernel(const void* __restrict__ labels, int label_bytes, float* __restrict__ mask, int B,
                                                        int D, int H, int W, int kd, int kh, int kw) {
    const int Do = D / kd, Ho = H / kh, Wo = W / kw;
    const long long total = (long long)B * Do * Ho * Wo;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= total) return;
    const int lane = threadIdx.x & 63;
    long long q = i;
    const int x = (int)(q % Wo); q /= Wo;
    const int y = (int)(q % Ho); q /= Ho;
    const int z = (int)(q % Do);
    const int b = (int)(q / Do);
    const int cell = kd * kh * kw;
    float s = 0.f;
    for (int e = lane; e < cell; e += 64) {
        const int dx = e % kw, dy = (e / kw) % kh, dz = e / (kw * kh);
        s += (float)load_label(labels, label_bytes, (((long long)b * D + z * kd + dz) * H + y * kh + dy) * W + x * kw + dx);
    }
    s = wave_sum(s);      // integer-valued partial sums: exact in any order
    if (lane == 0) mask[i] = (s / (float)cell) > 0.5f ? 1.f : 0.f;
}

// =================================================================================================
// FeCL, blockwise.  One workgroup = 64 rows (patches) of one sample; it walks all 64-column tiles,
// recomputing the Gram tile S = F_I F_J^T on the matrix cores (f32 MFMA: the epilogue's exp/log
// chain needs fp32 logits) from LDS-staged rows.  Nothing of size N x N ever reaches HBM.
//
// Restated semantics (utils/dycon_losses.py:172-234), per sample:
//   L_ij = S_ij/tau (i != j), L_ii = 0;  m_j = max_i L_ij (column max; == row max by symmetry);
//   a_ij = exp(L_ij - m_j);  n_i = sum_{k: class differs} a_ik;  P_ij = a_ij/(a_ij + n_i + 1e-18);
//   loss_i = u_i * sum_{j same class, j != i} -log(P_ij+1e-18) * w_ij / (cnt_i - 1 + 1e-18),
//   w_ij = (1-P_ij)^gamma (focal, differentiable) or 1;  cross: -log(1 - f_i.t_j + 1e-18) on
//   different-class pairs with f_i.t_j > thr, summed over the batch / their count.
// =================================================================================================
constexpr int FT = 64;  // tile edge
#ifndef FECL_P4_PREF
#define FECL_P4_PREF 0
#endif

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr_t;

// Tile geometry per storage type.  fp32 (parity mode): fp32 rows in LDS, v_mfma_f32_16x16x4_f32 (exact fp32 products).
// bf16: bf16 rows in LDS, v_mfma_f32_16x16x32_bf16 (exact products of the stored values, fp32 accumulate) -- 8x fewer
// matrix-core cycles and half the LDS bytes; the epilogue (exp/log chain) is fp32 in both.
template <typename T> struct FeclTile;
template <> struct FeclTile<float> {
    typedef float E;
    static constexpr int KS = 16, PAD = 4, TPAD = 4;
};
template <> struct FeclTile<bf16> {
    typedef unsigned short E;
    static constexpr int KS = 32, PAD = 16, TPAD = 8;   // row stride = 2 (mod 16) 16-byte slots: every ds_read_b128 lane group of a Gram fragment
                                                        // (rows r, slot offset kg) hits 16 distinct slots; PAD = 8 had 41 % of the LDS cycles as conflicts
};

__device__ __forceinline__ void stage_rows(float* __restrict__ dst, int stride, int Dp, const float* __restrict__ src, int row0,
                                           int N, int Dm) {
    for (int e = threadIdx.x; e < FT * Dp; e += 256) {
        const int r = e / Dp, k = e - r * Dp;
        float v = 0.f;
        if (row0 + r < N && k < Dm) v = src[(long long)(row0 + r) * Dm + k];
        dst[r * stride + k] = v;
    }
}
__device__ __forceinline__ void stage_rows(unsigned short* __restrict__ dst, int stride, int Dp, const bf16* __restrict__ src,
                                           int row0, int N, int Dm) {
    const unsigned short* s16 = reinterpret_cast<const unsigned short*>(src);
    if ((Dm & 7) == 0 && 256 % (Dp >> 3) == 0) {   // 16-byte pieces, a thread's column fixed: no division per piece (see fecl_kernel)
        const int ppr = Dp >> 3, rstep = 256 / ppr;
        const int rr0 = threadIdx.x / ppr, k = (threadIdx.x - rr0 * ppr) << 3;
        for (int r = rr0; r < FT; r += rstep) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row0 + r < N && k < Dm) v = *reinterpret_cast<const uint4*>(s16 + (long long)(row0 + r) * Dm + k);
            *reinterpret_cast<uint4*>(dst + r * stride + k) = v;
        }
    } else if ((Dm & 7) == 0) {   // 16-byte pieces
        const int ppr = Dp >> 3;
        for (int e = threadIdx.x; e < FT * ppr; e += 256) {
            const int r = e / ppr, k = (e - r * ppr) << 3;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row0 + r < N && k < Dm) v = *reinterpret_cast<const uint4*>(s16 + (long long)(row0 + r) * Dm + k);
            *reinterpret_cast<uint4*>(dst + r * stride + k) = v;
        }
    } else {
        for (int e = threadIdx.x; e < FT * Dp; e += 256) {
            const int r = e / Dp, k = e - r * Dp;
            dst[r * stride + k] = (row0 + r < N && k < Dm) ? s16[(long long)(row0 + r) * Dm + k] : (unsigned short)0;
        }
    }
}

// S tile for this wave: rows 16*wave..+15 of Fi against the 64 rows of Fj (both row-major, K contiguous).
__device__ __forceinline__ void gram_tile(const float* __restrict__ Fi, const float* __restrict__ Fj, int stride, int nq,
                                          int wave, int lane, f32x4 acc[4]) {
    const int r = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ap = Fi + (16 * wave + r) * stride + 4 * kg;
    const float* bp = Fj + r * stride + 4 * kg;
    for (int q = 0; q < nq; ++q) {   // K order inside a 16-chunk: k = 16q + 4*kg + e  (one ds_read_b128 per lane)
        const float4 a = *reinterpret_cast<const float4*>(ap + 16 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 b = *reinterpret_cast<const float4*>(bp + 16 * j * stride + 16 * q);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[j], 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void gram_tile(const unsigned short* __restrict__ Fi, const unsigned short* __restrict__ Fj, int stride,
                                          int nq, int wave, int lane, f32x4 acc[4]) {
    const int r = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned short* ap = Fi + (16 * wave + r) * stride + 8 * kg;
    const unsigned short* bp = Fj + r * stride + 8 * kg;
    for (int q = 0; q < nq; ++q) {   // k = 32q + 8*kg + e
        const bf16x8_t a = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ap + 32 * q));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8_t b = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(bp + 16 * j * stride + 32 * q));
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        }
    }
}

// gf[16 rows of this wave][16*t + .] += W[rows][64] * Fj[64][.]   (W = the per-pair weight tile, A operand; Fj = B operand)
template <int MAXNT>
__device__ __forceinline__ void weight_gemm(const float* __restrict__ Tt, int tstride, const float* __restrict__ Fj, int stride,
                                            int ntile, int wave, int lane, f32x4 gacc[MAXNT]) {
    const int r = lane & 15, kg = lane >> 4;
    for (int s = 0; s < FT / 4; ++s) {
        const float a = Tt[(16 * wave + r) * tstride + 4 * s + kg];
        const float* bp = Fj + (4 * s + kg) * stride + r;
#pragma unroll
        for (int t = 0; t < MAXNT; ++t)
            if (t < ntile) gacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[16 * t], gacc[t], 0, 0, 0);
    }
}
template <int MAXNT>
__device__ __forceinline__ void weight_gemm(const unsigned short* __restrict__ Tt, int tstride, const unsigned short* __restrict__ Fj,
                                            int stride, int ntile, int wave, int lane, f32x4 gacc[MAXNT]) {
    const int r = lane & 15, kg = lane >> 4, q = r >> 2, p = r & 3;
#pragma unroll
    for (int s = 0; s < FT / 32; ++s) {
        const bf16x8_t a = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Tt + (16 * wave + r) * tstride + 32 * s + 8 * kg));
        // B[k = pair column j][col = feature d] = Fj[j][d]: 8 rows of Fj at one column -> transposed LDS read
        const unsigned short* b0 = Fj + (32 * s + 8 * kg + q) * stride + 4 * p;
#pragma unroll
        for (int t = 0; t < MAXNT; ++t) {
            if (t < ntile) {
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(b0 + 16 * t));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(b0 + 16 * t + 4 * stride));
                const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                gacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8_t, v), gacc[t], 0, 0, 0);
            }
        }
    }
}
__device__ __forceinline__ void put_weight(float* Tt, int idx, float v) { Tt[idx] = v; }
__device__ __forceinline__ void put_weight(unsigned short* Tt, int idx, float v) { Tt[idx] = f32_to_bf16_bits(v); }

// sum / max over the 16 lanes that share a row group (lane bits 0..3)
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64));
    v = fmaxf(v, __shfl_xor(v, 4, 64)); v = fmaxf(v, __shfl_xor(v, 8, 64));
    return v;
}

// fp32 storage (parity mode): IEEE expf/logf/division.  bf16 storage: fexp / flog / fdiv<true> (hardware forms, see the top of this file);
// every argument of the pair epilogue is in the normal range: exponents L - m in [-2/tau, 0], denominators a + n >= exp(-2/tau),
// 1 - s + 1e-18 >= 1e-18.

// d/dP of phi(P) = -log(P+eps) * (1-P)^gamma   (gamma = 0 <=> no focal weight)
template <bool FAST>
__device__ __forceinline__ void focal_terms(float P, float gamma, int focal, float& phi, float& dphi) {
    const float eps = 1e-18f;
    const float lg = flog<FAST>(P + eps);
    if (!focal) { phi = -lg; dphi = -fdiv<FAST>(1.f, P + eps); return; }
    const float om = fmaxf(1.f - P, 0.f);
    const float w = gamma == 2.f ? om * om : powf(om, gamma);
    const float dw = gamma == 2.f ? 2.f * om : (om > 0.f ? gamma * powf(om, gamma - 1.f) : 0.f);
    phi = -lg * w;
    dphi = -fdiv<FAST>(w, P + eps) + dw * lg;
}

// bf16 storage, focal gamma = 2 (the configuration every script runs): the same-class pair terms with the divisions folded away.
// With a = exp(L - m), d = a + n, r = 1/d:  P = a r, 1 - P = n r, log P = (L - m) - log d, and
//   phi(P)            = -log P (1 - P)^2
//   a * dphi/dP       = -(1-P)^2 d + 2 (1-P) a log P = n r (2 a log P - n)
// so one exp, one rcp and one log per direction replace one exp, one log and four divisions (the 1e-18 guards of the reference
// formula only matter for d < 1e-15, which a + n never is: n counts at least exp(-2/tau) per other-class column).
__device__ __forceinline__ void fast_pair(float lm /* L - m */, float n, float& a, float& r, float& lgP, float& adphi) {
    a = fexp<true>(lm);
    const float d = a + n;
    r = __builtin_amdgcn_rcpf(d);
    lgP = lm - flog<true>(d);
    adphi = n * r * (2.f * a * lgP - n);
}

// PASS: 1 row max | 2 negative sums + class counts | 3 loss + H (+ cross sums) | 4 gradient
template <typename T, int PASS>
__global__ __launch_bounds__(256) void fecl_kernel(const T* __restrict__ F, const T* __restrict__ Tch,
                                                   const float* __restrict__ mask, const float* __restrict__ gamb, int N, int Dm,
                                                   float tau, float gamma, int focal, float thr, float* __restrict__ ws,
                                                   double* __restrict__ out, const float* __restrict__ coef, float lambda_cross,
                                                   float* __restrict__ GS, int Btot, int CS, int tiles_per_split) {
    // blockIdx.z = column split: this workgroup walks column tiles [z*tiles_per_split, (z+1)*tiles_per_split) and writes
    // its PARTIAL row results into slab z; consumers combine the CS slabs on load (max / ordered sum): deterministic.
    typedef typename FeclTile<T>::E E;
    constexpr int KS = FeclTile<T>::KS;
    constexpr bool FAST = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    E* lds = reinterpret_cast<E*>(lds_raw);
    const int Dp = (Dm + KS - 1) / KS * KS, stride = Dp + FeclTile<T>::PAD, nq = Dp / KS;
    E* Fi = lds;
    E* Fj = lds + FT * stride;
    E* Tt = Fj + FT * stride;                          // PASS 4 only: 64 x 64 pair-weight tile
    constexpr int tstride = FT + FeclTile<T>::TPAD;
    // per-column statistics of the current column tile {max, negative sum, kappa, H, mask}[64]: loaded (and the column-split
    // slabs combined) ONCE per tile by 64 threads instead of by every lane for each of its four columns (at N = 15680 the
    // per-lane global loads were what the pair epilogue waited on)
    float* cst = reinterpret_cast<float*>(lds_raw + (((size_t)2 * FT * stride + (PASS == 4 ? FT * tstride : 0)) * sizeof(E) + 15) / 16 * 16);
    const int b = blockIdx.y, i0 = blockIdx.x * FT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const long long BN = (long long)Btot * N;
    const int cs = blockIdx.z;
    float* wm = ws;                          // [CS][BN] partial column max
    float* wn = ws + (long long)CS * BN;     // [CS][BN] partial negative sums
    float* wc = ws + 2LL * CS * BN;          // [CS][BN] partial same-class counts
    float* wh = ws + 3LL * CS * BN;          // [CS][BN] partial kappa_i * H_i
    float* wk = ws + 4LL * CS * BN;          // [BN]     kappa_i = u_i / (cnt_i - 1 + 1e-18)
    auto ld_max = [&](const float* base, long long idx) { float v = base[idx]; for (int z = 1; z < CS; ++z) v = fmaxf(v, base[z * BN + idx]); return v; };
    auto ld_sum = [&](const float* base, long long idx) { float v = base[idx]; for (int z = 1; z < CS; ++z) v += base[z * BN + idx]; return v; };
    const T* Fb = F + (long long)b * N * Dm;
    const T* Tb = Tch ? Tch + (long long)b * N * Dm : nullptr;
    const float* mb = mask + (long long)b * N;
    const long long rb = (long long)b * N;

    stage_rows(Fi, stride, Dp, Fb, i0, N, Dm);

    // per-lane row data: rows gi[i] = i0 + 16*wave + 4*kg + i
    int gi[4];
    float mrow[4], nrow[4], krow[4], hrow[4], mi[4];
    bool vi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        gi[i] = i0 + 16 * wave + 4 * kg + i;
        vi[i] = gi[i] < N;
        mrow[i] = vi[i] ? mb[gi[i]] : -1.f;
        mi[i] = nrow[i] = krow[i] = hrow[i] = 0.f;
        if (PASS >= 2 && vi[i]) mi[i] = ld_max(wm, rb + gi[i]);
        if (PASS >= 3 && vi[i]) nrow[i] = ld_sum(wn, rb + gi[i]);
        if (PASS >= 4 && vi[i]) { krow[i] = wk[rb + gi[i]]; hrow[i] = ld_sum(wh, rb + gi[i]); }
    }
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};   // per-row accumulators of the pass
    float cnum = 0.f, ccnt = 0.f;

    // PASS 4: persistent output tile  gf[16 rows of this wave][Dp], as Dp/16 accumulators
    constexpr int MAXNT = 16;   // Dm <= 256
    const int ntile = (Dm + 15) / 16;
    f32x4 gacc[MAXNT];
    if (PASS == 4) {
#pragma unroll
        for (int t = 0; t < MAXNT; ++t) gacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float inv_tau = 1.f / tau;
    float cross_scale = 0.f, stud_scale = 0.f;
    if (PASS == 4) {
        stud_scale = coef[0] / (float)BN;
        cross_scale = Tb ? coef[0] * lambda_cross / ((float)out[2] + 1e-18f) : 0.f;
    }

    const int j_beg = cs * tiles_per_split * FT, j_end = min(N, (cs + 1) * tiles_per_split * FT);
    // bf16 rows in 16-byte pieces: the NEXT tile to be staged (the teacher rows of this column tile, or the student rows of the
    // next one) is requested into registers before the current tile's MFMAs / pair epilogue and written to LDS behind the barrier
    // that retires it -- the global round trip no longer sits between two barriers with nothing else to do.
    constexpr int NPF = 8;                                // 64 rows x 256 bf16 = 2048 pieces / 256 threads
    // (not in the gradient pass: its 64 accumulator registers plus a staged tile leave room for one wave per SIMD only -- measured slower)
    const bool pref = (PASS != 4 || FECL_P4_PREF) && sizeof(T) == 2 && (Dm & 7) == 0 && Dp <= 256;
    uint4 pfr[NPF];
    // A thread's piece of a staged tile: with 256 % (pieces per row) == 0 (Dp = 32 .. 256 in powers of two) its column never changes
    // and its row advances by a fixed step, so nothing is divided per piece -- the e / ppr form below cost ~25 vector instructions per
    // piece, 16 pieces per tile: 400 of the 468 VALU instructions per tile and wave that PMC counted in pass 1
    // (profiles/r03_fecl_pmc.txt), with the vector ALU the busiest pipe of passes 1 and 2.
    const int ppr_ = Dp >> 3;
    const bool lin = 256 % ppr_ == 0;
    const int rstep = lin ? 256 / ppr_ : 0, rr0 = threadIdx.x / ppr_, k0 = (threadIdx.x - rr0 * ppr_) << 3;
    auto prefetch = [&](const T* src, int row0) {
        const unsigned short* s16 = reinterpret_cast<const unsigned short*>(src);
        const int ppr = Dp >> 3;
        if (lin) {
#pragma unroll
            for (int it = 0; it < NPF; ++it) {
                const int rr = rr0 + it * rstep;
                pfr[it] = make_uint4(0, 0, 0, 0);
                if (rr < FT && row0 + rr < N && k0 < Dm) pfr[it] = *reinterpret_cast<const uint4*>(s16 + (long long)(row0 + rr) * Dm + k0);
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NPF; ++it) {
            const int e = threadIdx.x + 256 * it;
            const int rr = e / ppr, k = (e - rr * ppr) << 3;
            pfr[it] = make_uint4(0, 0, 0, 0);
            if (e < FT * ppr && row0 + rr < N && k < Dm) pfr[it] = *reinterpret_cast<const uint4*>(s16 + (long long)(row0 + rr) * Dm + k);
        }
    };
    auto commit = [&]() {
        const int ppr = Dp >> 3;
        if (lin) {
#pragma unroll
            for (int it = 0; it < NPF; ++it) {
                const int rr = rr0 + it * rstep;
                if (rr < FT) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(Fj) + rr * stride + k0) = pfr[it];
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NPF; ++it) {
            const int e = threadIdx.x + 256 * it;
            const int rr = e / ppr, k = (e - rr * ppr) << 3;
            if (e < FT * ppr) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(Fj) + rr * stride + k) = pfr[it];
        }
    };
    const bool has_cross = (PASS == 3 || PASS == 4) && Tb;
    // Column statistics of a tile (the slabs of the column splits combined) are requested one tile AHEAD, into registers of the first
    // 64 threads, and written to LDS behind the barrier that retires the previous tile: their global round trip (up to 3 x CS + 2
    // loads per column) used to sit exposed between the two barriers of every tile -- at N = 15 680 that, not the matrix cores or
    // the pair epilogue, set the time of passes 2-4.
    float pst[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    auto prefetch_stats = [&](int jt) {
        if (PASS >= 2 && threadIdx.x < FT) {
            const int gc = jt + threadIdx.x;
            const bool vc = gc < N;
            pst[0] = vc ? ld_max(wm, rb + gc) : 0.f;
            pst[4] = vc ? mb[gc] : -2.f;
            if (PASS == 4) {
                pst[1] = vc ? ld_sum(wn, rb + gc) : 0.f;
                pst[2] = vc ? wk[rb + gc] : 0.f;
                pst[3] = vc ? ld_sum(wh, rb + gc) : 0.f;
            }
        }
    };
    if (pref && j_beg < j_end) prefetch(Fb, j_beg);
    if (j_beg < j_end) prefetch_stats(j_beg);
    for (int j0 = j_beg; j0 < j_end; j0 += FT) {
        __syncthreads();
        if (pref) commit(); else stage_rows(Fj, stride, Dp, Fb, j0, N, Dm);
        if (PASS >= 2 && threadIdx.x < FT) {
            cst[threadIdx.x] = pst[0];
            cst[4 * FT + threadIdx.x] = pst[4];
            if (PASS == 4) {
                cst[FT + threadIdx.x] = pst[1];
                cst[2 * FT + threadIdx.x] = pst[2];
                cst[3 * FT + threadIdx.x] = pst[3];
            }
        }
        __syncthreads();
        if (pref) {
            if (has_cross) prefetch(Tb, j0);
            else if (j0 + FT < j_end) prefetch(Fb, j0 + FT);
        }
        if (j0 + FT < j_end) prefetch_stats(j0 + FT);
        f32x4 acc[4];
        gram_tile(Fi, Fj, stride, nq, wave, lane, acc);

        if (PASS == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gj = j0 + 16 * j + r;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (gj < N && gj != gi[i]) a0[i] = fmaxf(a0[i], FAST ? acc[j][i] * inv_tau : acc[j][i] / tau);
            }
        } else if (PASS == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gj = j0 + 16 * j + r;
                if (gj >= N) continue;
                const float mj = cst[16 * j + r], mk = cst[4 * FT + 16 * j + r];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (mk == mrow[i]) a1[i] += 1.f;
                    else a0[i] += FAST ? fexp<true>(acc[j][i] * inv_tau - mj) : expf(acc[j][i] / tau - mj);
                }
            }
        } else if (PASS == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gj = j0 + 16 * j + r;
                if (gj >= N) continue;
                const float mj = cst[16 * j + r], mk = cst[4 * FT + 16 * j + r];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (mk == mrow[i]) {
                        a1[i] += 1.f;
                        if (gj != gi[i] && FAST && focal && gamma == 2.f) {
                            float a, rr, lgP, adphi;
                            fast_pair(acc[j][i] * inv_tau - mj, nrow[i], a, rr, lgP, adphi);
                            const float om = nrow[i] * rr;
                            a0[i] += -lgP * om * om;                      // phi
                            hrow[i] -= adphi * rr * rr;                   // dphi * (-a / d^2)
                        } else if (gj != gi[i]) {
                            const float a = fexp<FAST>(fdiv<FAST>(acc[j][i], tau) - mj);
                            const float den = a + nrow[i] + 1e-18f;
                            const float P = fdiv<FAST>(a, den);
                            float phi, dphi;
                            focal_terms<FAST>(P, gamma, focal, phi, dphi);
                            a0[i] += phi;
                            hrow[i] += dphi * (-fdiv<FAST>(a, den * den));
                        }
                    }
                }
            }
        } else {  // PASS 4: W_ij = dL_ij + dL_ji  -> LDS
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gj = j0 + 16 * j + r;
                const bool vj = gj < N;
                const int cc = 16 * j + r;
                const float mj = cst[cc], mk = cst[4 * FT + cc], nj = cst[FT + cc], kj = cst[2 * FT + cc], hj = cst[3 * FT + cc];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float tv = 0.f;
                    if (FAST && focal && gamma == 2.f && vj && vi[i] && gj != gi[i]) {
                        const float l = acc[j][i] * inv_tau;
                        if (mk == mrow[i]) {     // tv = a kappa dphi n / d^2 for both directions
                            float a, rr, lgP, adphi;
                            fast_pair(l - mj, nrow[i], a, rr, lgP, adphi);
                            tv = krow[i] * nrow[i] * rr * rr * adphi;
                            fast_pair(l - mi[i], nj, a, rr, lgP, adphi);
                            tv += kj * nj * rr * rr * adphi;
                        } else {
                            tv = fexp<true>(l - mj) * hrow[i] + fexp<true>(l - mi[i]) * hj;
                        }
                        tv *= stud_scale * inv_tau;
                    } else if (vj && vi[i] && gj != gi[i]) {
                        const float l = fdiv<FAST>(acc[j][i], tau);
                        const float aij = fexp<FAST>(l - mj), aji = fexp<FAST>(l - mi[i]);
                        if (mk == mrow[i]) {
                            float phi, dphi;
                            const float dij = aij + nrow[i] + 1e-18f;
                            focal_terms<FAST>(fdiv<FAST>(aij, dij), gamma, focal, phi, dphi);
                            tv = fdiv<FAST>(aij * krow[i] * dphi * (nrow[i] + 1e-18f), dij * dij);
                            const float dji = aji + nj + 1e-18f;
                            focal_terms<FAST>(fdiv<FAST>(aji, dji), gamma, focal, phi, dphi);
                            tv += fdiv<FAST>(aji * kj * dphi * (nj + 1e-18f), dji * dji);
                        } else {
                            tv = aij * hrow[i] + aji * hj;
                        }
                        tv *= stud_scale / tau;   // (uniform scalar)
                    }
                    put_weight(Tt, (16 * wave + 4 * kg + i) * tstride + 16 * j + r, tv);
                }
            }
            // rows 16*wave .. 16*wave+15 of the pair-weight tile are written AND read by this wave only (LDS operations of one wave
            // execute in order): no workgroup barrier between the epilogue and the gradient GEMM
            __builtin_amdgcn_wave_barrier();
            weight_gemm<MAXNT>(Tt, tstride, Fj, stride, ntile, wave, lane, gacc);
        }

        if ((PASS == 3 || PASS == 4) && Tb) {   // cross branch against the teacher rows of this column tile
            __syncthreads();
            if (pref) commit(); else stage_rows(Fj, stride, Dp, Tb, j0, N, Dm);
            __syncthreads();
            if (pref && j0 + FT < j_end) prefetch(Fb, j0 + FT);
            gram_tile(Fi, Fj, stride, nq, wave, lane, acc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gj = j0 + 16 * j + r;
                const bool vj = gj < N;
                const float mk = cst[4 * FT + 16 * j + r];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sx = acc[j][i];
                    const bool hard = vj && vi[i] && mk != mrow[i] && sx > thr;
                    if (PASS == 3) {
                        if (hard) { cnum += -flog<FAST>(1.f - sx + 1e-18f); ccnt += 1.f; }
                    } else {
                        put_weight(Tt, (16 * wave + 4 * kg + i) * tstride + 16 * j + r, hard ? fdiv<FAST>(cross_scale, 1.f - sx + 1e-18f) : 0.f);
                    }
                }
            }
            if (PASS == 4) {
                __builtin_amdgcn_wave_barrier();
                weight_gemm<MAXNT>(Tt, tstride, Fj, stride, ntile, wave, lane, gacc);
            }
        }
    }

    // ---- per-pass finish
    if (PASS == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = row16_max(a0[i]);
            if (r == 0 && vi[i]) wm[cs * BN + rb + gi[i]] = v;
        }
    } else if (PASS == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ns = row16_sum(a0[i]), cs_ = row16_sum(a1[i]);
            if (r == 0 && vi[i]) { wn[cs * BN + rb + gi[i]] = ns; wc[cs * BN + rb + gi[i]] = cs_; }
        }
    } else if (PASS == 3) {
        float lsum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ph = row16_sum(a0[i]), hs = row16_sum(hrow[i]);
            if (r == 0 && vi[i]) {
                const float u = gamb ? gamb[rb + gi[i]] : 1.f;
                const float kap = u / (ld_sum(wc, rb + gi[i]) - 1.f + 1e-18f);   // total same-class count from pass 2
                if (cs == 0) wk[rb + gi[i]] = kap;
                wh[cs * BN + rb + gi[i]] = kap * hs;
                lsum += ph * kap;
            }
        }
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds_raw);   // staging buffers are dead; ALL LDS stays in the one dynamic array
        const float bs = block_sum(lsum, red), bn = block_sum(cnum, red), bc = block_sum(ccnt, red);
        if (threadIdx.x == 0) {
            atomicAdd(&out[0], (double)bs);
            if (Tb) { atomicAdd(&out[1], (double)bn); atomicAdd(&out[2], (double)bc); }
        }
    } else {
        float* gb = GS + ((long long)cs * Btot + b) * N * Dm;   // fp32 slab of this column split
#pragma unroll
        for (int t = 0; t < MAXNT; ++t) {
            if (t >= ntile) continue;
            const int d = 16 * t + r;
            if (d >= Dm) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (vi[i]) gb[(long long)gi[i] * Dm + d] = gacc[t][i];
        }
    }
}

// =================================================================================================
// FeCL passes 1-3 at large N (bf16 storage; focal gamma = 2, or no focal weight): 128-row blocks on v_mfma_f32_32x32x16_bf16.
//
// fecl_kernel above gives a workgroup 64 rows and re-reads, per 64-column tile, both operands from LDS (160 KB of ds_read_b128 per
// tile for 64 x 64 pairs), stages the column tile between TWO barriers and fetches it once per 64 rows: at N = 15 680 its passes run
// at 13-18 % of the matrix pipe with the waves waiting (barriers, LDS) half their life (profiles/r03_fecl_n15680.txt).  Here
//   * a workgroup owns 128 rows, a wave 32 of them; the wave's row fragments (B operand: 32 rows x Dm, 64 registers) are loaded
//     from global memory ONCE and stay in registers for all column tiles;
//   * the column tile is the A operand (32 columns x 16 k per instruction, one ds_read_b128 each), so the product comes out
//     transposed: a lane holds ONE row i (its B column) against 32 columns j of the tile, the row's state and accumulators are
//     scalars per lane, and the per-column statistics are 16-byte LDS broadcasts;
//   * column tiles are double-buffered in LDS: one barrier per tile, the next tile's global loads in registers meanwhile;
//   * 64 KB of LDS reads per tile and wave for 32 x 64 pairs (2.5 x fewer per pair), a tile fetched once per 128 rows.
// Same arithmetic per pair as fecl_kernel<bf16> (fast_pair / hardware exp, log, rcp), same workspace layout: the gradient pass
// (fecl_kernel<bf16, 4>) consumes what these passes leave.  Row sums are accumulated in a different order than fecl_kernel's
// (fp32, ~1e-7 relative).
// =================================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int F2_ROWS = 128;   // rows per workgroup (4 waves x 32)
constexpr int F2_PAD = 8;      // column-tile row stride Dm + 8 elements = odd number of 16-byte slots: the 32 rows of an A-fragment read
                               // (lanes 0-31 slot 2q, lanes 32-63 slot 2q + 1) fall in 16 distinct slots per ds_read_b128 lane group

template <int PASS, int NQ>     // NQ = Dm / 16 k-steps (compile time: a predicated k-loop made the compiler copy the 32 accumulators per step)
__global__ __launch_bounds__(256, 2) void fecl_rows128_kernel(const bf16* __restrict__ F, const bf16* __restrict__ Tch,
                                                              const float* __restrict__ mask, const float* __restrict__ gamb, int N,
                                                              int Dm, float tau, int focal, float thr, float* __restrict__ ws,
                                                              double* __restrict__ out, int Btot, int CS, int tiles_per_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int stride = Dm + F2_PAD, tile_elems = FT * stride;
    unsigned short* Fj = reinterpret_cast<unsigned short*>(lds_raw);                       // [2][64][stride]
    float* cst = reinterpret_cast<float*>(lds_raw + ((size_t)2 * tile_elems * 2 + 15) / 16 * 16);   // [2][{max, mask}][64]
    const int b = blockIdx.y, cs = blockIdx.z, i0 = blockIdx.x * F2_ROWS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rl = lane & 31, half = lane >> 5;
    const long long BN = (long long)Btot * N, rb = (long long)b * N;
    float* wm = ws;
    float* wn = ws + (long long)CS * BN;
    float* wc = ws + 2LL * CS * BN;
    float* wh = ws + 3LL * CS * BN;
    float* wk = ws + 4LL * CS * BN;
    auto ld_max = [&](const float* base, long long idx) { float v = base[idx]; for (int z = 1; z < CS; ++z) v = fmaxf(v, base[z * BN + idx]); return v; };
    auto ld_sum = [&](const float* base, long long idx) { float v = base[idx]; for (int z = 1; z < CS; ++z) v += base[z * BN + idx]; return v; };
    const unsigned short* Fb = reinterpret_cast<const unsigned short*>(F) + (long long)b * N * Dm;
    const unsigned short* Tb = Tch ? reinterpret_cast<const unsigned short*>(Tch) + (long long)b * N * Dm : nullptr;
    const float* mb = mask + rb;

    // this lane's row: fragments (k = 16 q + 8 half .. + 7) resident for the whole kernel
    const int gi = i0 + 32 * wave + rl;
    const bool vi = gi < N;
    uint4 bfrag[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        bfrag[q] = make_uint4(0, 0, 0, 0);
        if (vi) bfrag[q] = *reinterpret_cast<const uint4*>(Fb + (long long)gi * Dm + 16 * q + 8 * half);
    }
    const float mrow = vi ? mb[gi] : -1.f;
    float nrow = 0.f;
    if (PASS >= 3 && vi) nrow = ld_sum(wn, rb + gi);
    float a0 = 0.f, a1 = 0.f, hrow = 0.f, cnum = 0.f, ccnt = 0.f;
    const float inv_tau = 1.f / tau;

    // staging: a thread owns one 16-byte column of the tile and every rstep-th row (256 % (Dm / 8) == 0: checked on the host)
    const int ppr = Dm >> 3, rstep = 256 / ppr, rr0 = threadIdx.x / ppr, k0 = (threadIdx.x - rr0 * ppr) << 3;
    constexpr int NPF = 8;
    uint4 pfr[NPF];
    float pst[2] = {0.f, -2.f};
    const bool cross = PASS == 3 && Tb != nullptr;
    const int j_beg = cs * tiles_per_split * FT, j_end = min(N, (cs + 1) * tiles_per_split * FT);
    const int ntile = j_end > j_beg ? (j_end - j_beg + FT - 1) / FT : 0;
    const int nst = cross ? 2 * ntile : ntile;                 // staged tiles: F_0 (, T_0), F_1 (, T_1), ...
    auto col0 = [&](int s_) { return j_beg + (cross ? s_ >> 1 : s_) * FT; };
    auto prefetch = [&](int s_) {
        const unsigned short* src = cross && (s_ & 1) ? Tb : Fb;
        const int row0 = col0(s_);
#pragma unroll
        for (int it = 0; it < NPF; ++it) {
            const int rr = rr0 + it * rstep;
            pfr[it] = make_uint4(0, 0, 0, 0);
            if (rr < FT && row0 + rr < N) pfr[it] = *reinterpret_cast<const uint4*>(src + (long long)(row0 + rr) * Dm + k0);
        }
        if (threadIdx.x < FT) {
            const int gc = row0 + threadIdx.x;
            pst[0] = PASS >= 2 && gc < N ? ld_max(wm, rb + gc) : 0.f;
            pst[1] = gc < N ? mb[gc] : -2.f;
        }
    };
    auto commit = [&](int buf) {
        unsigned short* dst = Fj + buf * tile_elems;
#pragma unroll
        for (int it = 0; it < NPF; ++it) {
            const int rr = rr0 + it * rstep;
            if (rr < FT) *reinterpret_cast<uint4*>(dst + rr * stride + k0) = pfr[it];
        }
        if (threadIdx.x < FT) { cst[buf * 2 * FT + threadIdx.x] = pst[0]; cst[buf * 2 * FT + FT + threadIdx.x] = pst[1]; }
    };

    if (nst > 0) { prefetch(0); commit(0); }
    if (nst > 1) prefetch(1);
    __syncthreads();
    for (int s_ = 0; s_ < nst; ++s_) {
        const int buf = s_ & 1, j0 = col0(s_);
        const unsigned short* At = Fj + buf * tile_elems + rl * stride + 8 * half;
        f32x16 acc[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[jt][e] = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const bf16x8_t a = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(At + jt * 32 * stride + 16 * q));
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, bfrag[q]), acc[jt], 0, 0, 0);
            }
        }
        // D[m = column of the tile][n = this lane's row]: register 4 g + e of acc[jt] is column jt*32 + 8 g + 4 half + e
        const float* cmx = cst + buf * 2 * FT;
        const bool teacher_tile = cross && (s_ & 1);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int jl = jt * 32 + 8 * g + 4 * half;
                const float4 m4 = *reinterpret_cast<const float4*>(cmx + jl);
                const float4 k4 = *reinterpret_cast<const float4*>(cmx + FT + jl);
                const float mj4[4] = {m4.x, m4.y, m4.z, m4.w}, mk4[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int gj = j0 + jl + e;
                    const float v = acc[jt][4 * g + e];
                    if (gj >= N) continue;
                    if (PASS == 1) {
                        if (gj != gi) a0 = fmaxf(a0, v * inv_tau);
                    } else if (PASS == 2) {                      // branch-free (a divergent if / else walks both paths anyway)
                        const bool same = mk4[e] == mrow;
                        const float ex = fexp<true>(v * inv_tau - mj4[e]);
                        a1 += same ? 1.f : 0.f;
                        a0 += same ? 0.f : ex;
                    } else if (!teacher_tile) {
                        const bool same = mk4[e] == mrow;
                        a1 += same ? 1.f : 0.f;
                        float a, rr, lgP, adphi;
                        fast_pair(v * inv_tau - mj4[e], nrow, a, rr, lgP, adphi);
                        const float om = nrow * rr;
                        const float phi = focal ? -lgP * om * om : -lgP;            // phi(P); without focal: -log P
                        const float dh = focal ? -adphi * rr * rr : rr;              // dphi * (-a / d^2); without focal: 1 / d
                        const bool on = same && gj != gi;
                        a0 += on ? phi : 0.f;
                        hrow += on ? dh : 0.f;
                    } else {
                        if (vi && mk4[e] != mrow && v > thr) { cnum += -flog<true>(1.f - v + 1e-18f); ccnt += 1.f; }
                    }
                }
            }
        if (s_ + 1 < nst) commit(buf ^ 1);
        if (s_ + 2 < nst) prefetch(s_ + 2);
        __syncthreads();
    }

    // the two halves of a wave hold the same 32 rows against different columns
    if (PASS == 1) {
        const float v = fmaxf(a0, __shfl_xor(a0, 32, 64));
        if (half == 0 && vi) wm[cs * BN + rb + gi] = v;
    } else if (PASS == 2) {
        const float ns = a0 + __shfl_xor(a0, 32, 64), cn = a1 + __shfl_xor(a1, 32, 64);
        if (half == 0 && vi) { wn[cs * BN + rb + gi] = ns; wc[cs * BN + rb + gi] = cn; }
    } else {
        const float ph = a0 + __shfl_xor(a0, 32, 64), hs = hrow + __shfl_xor(hrow, 32, 64);
        float lsum = 0.f;
        if (half == 0 && vi) {
            const float u = gamb ? gamb[rb + gi] : 1.f;
            const float kap = u / (ld_sum(wc, rb + gi) - 1.f + 1e-18f);   // total same-class count from pass 2
            if (cs == 0) wk[rb + gi] = kap;
            wh[cs * BN + rb + gi] = kap * hs;
            lsum = ph * kap;
        }
        float* red = reinterpret_cast<float*>(lds_raw);   // the tile buffers are dead (the loop ends on a barrier)
        const float bs = block_sum(lsum, red), bn = block_sum(cnum, red), bc = block_sum(ccnt, red);
        if (threadIdx.x == 0) {
            atomicAdd(&out[0], (double)bs);
            if (Tb) { atomicAdd(&out[1], (double)bn); atomicAdd(&out[2], (double)bc); }
        }
    }
}

// ---- the gradient pass on the same 128-row blocks.
// After the transposed Gram product a lane holds, for its row i, the pair weights W_ij of 16 columns per 32-column half tile:
// registers 4 g + e <-> column 8 g + 4 half + e.  The gradient GEMM gf^T[d][i] += sum_j F_J^T[d][j] W^T[j][i] contracts over j in ANY
// order, so k-step s of a half tile is DEFINED as the 16 columns {16 s + 8 g' + 4 half + e : g' in {0, 1}}: its B operand (W^T, 8 values
// per lane: k-slot 4 g' + e of this lane's half) is exactly registers 4 (2 s + g') + e of the lane's own accumulator -- converted to bf16
// and packed, no LDS round trip and no cross-lane traffic -- and its A operand (F_J^T: feature d = the lane's M index, the same 8
// columns) is two transposed LDS reads (ds_read_b64_tr_b16: 4 consecutive rows j of the tile at 16 features per 16-lane group).
// gf (32 rows x Dm per wave) stays in registers for the whole column sweep: 8 NQ accumulator registers; one wave per SIMD.
template <int NQ>
__global__ __launch_bounds__(256, 1) void fecl_rows128_grad_kernel(const bf16* __restrict__ F, const bf16* __restrict__ Tch,
                                                                   const float* __restrict__ mask, int N, int Dm, float tau, int focal,
                                                                   float thr, const float* __restrict__ ws, const double* __restrict__ out,
                                                                   const float* __restrict__ coef, float lambda_cross,
                                                                   float* __restrict__ GS, int Btot, int CS, int tiles_per_split) {
    constexpr int NDT = NQ / 2;                                   // 32-feature blocks of the gradient
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int stride = Dm + F2_PAD, tile_elems = FT * stride;
    unsigned short* Fj = reinterpret_cast<unsigned short*>(lds_raw);                       // [2][64][stride]
    float* cst = reinterpret_cast<float*>(lds_raw + ((size_t)2 * tile_elems * 2 + 15) / 16 * 16);   // [2][{max, mask, n, kappa, H}][64]
    const int b = blockIdx.y, cs = blockIdx.z, i0 = blockIdx.x * F2_ROWS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rl = lane & 31, half = lane >> 5;
    const long long BN = (long long)Btot * N, rb = (long long)b * N;
    const float* wm = ws;
    const float* wn = ws + (long long)CS * BN;
    const float* wh = ws + 3LL * CS * BN;
    const float* wk = ws + 4LL * CS * BN;
    auto ld_max = [&](const float* base, long long idx) { float v = base[idx]; for (int z = 1; z < CS; ++z) v = fmaxf(v, base[z * BN + idx]); return v; };
    auto ld_sum = [&](const float* base, long long idx) { float v = base[idx]; for (int z = 1; z < CS; ++z) v += base[z * BN + idx]; return v; };
    const unsigned short* Fb = reinterpret_cast<const unsigned short*>(F) + (long long)b * N * Dm;
    const unsigned short* Tb = Tch ? reinterpret_cast<const unsigned short*>(Tch) + (long long)b * N * Dm : nullptr;
    const float* mb = mask + rb;

    const int gi = i0 + 32 * wave + rl;
    const bool vi = gi < N;
    uint4 bfrag[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        bfrag[q] = make_uint4(0, 0, 0, 0);
        if (vi) bfrag[q] = *reinterpret_cast<const uint4*>(Fb + (long long)gi * Dm + 16 * q + 8 * half);
    }
    const float mrow = vi ? mb[gi] : -1.f;
    float mi = 0.f, nrow = 0.f, krow = 0.f, hrow = 0.f;
    if (vi) { mi = ld_max(wm, rb + gi); nrow = ld_sum(wn, rb + gi); krow = wk[rb + gi]; hrow = ld_sum(wh, rb + gi); }
    const float inv_tau = 1.f / tau;
    const float stud_scale = coef[0] / (float)BN * inv_tau;
    const float cross_scale = Tb ? coef[0] * lambda_cross / ((float)out[2] + 1e-18f) : 0.f;
    f32x16 gacc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) gacc[dt][e] = 0.f;

    const int ppr = Dm >> 3, rstep = 256 / ppr, rr0 = threadIdx.x / ppr, k0 = (threadIdx.x - rr0 * ppr) << 3;
    constexpr int NPF = 8;
    uint4 pfr[NPF];
    float pst[5] = {0.f, -2.f, 0.f, 0.f, 0.f};
    const bool cross = Tb != nullptr;
    const int j_beg = cs * tiles_per_split * FT, j_end = min(N, (cs + 1) * tiles_per_split * FT);
    const int ntile = j_end > j_beg ? (j_end - j_beg + FT - 1) / FT : 0;
    const int nst = cross ? 2 * ntile : ntile;
    auto col0 = [&](int s_) { return j_beg + (cross ? s_ >> 1 : s_) * FT; };
    auto prefetch = [&](int s_) {
        const bool tt = cross && (s_ & 1);
        const unsigned short* src = tt ? Tb : Fb;
        const int row0 = col0(s_);
#pragma unroll
        for (int it = 0; it < NPF; ++it) {
            const int rr = rr0 + it * rstep;
            pfr[it] = make_uint4(0, 0, 0, 0);
            if (rr < FT && row0 + rr < N) pfr[it] = *reinterpret_cast<const uint4*>(src + (long long)(row0 + rr) * Dm + k0);
        }
        if (threadIdx.x < FT) {
            const int gc = row0 + threadIdx.x;
            const bool vc = gc < N;
            pst[1] = vc ? mb[gc] : -2.f;
            if (!tt) {
                pst[0] = vc ? ld_max(wm, rb + gc) : 0.f;
                pst[2] = vc ? ld_sum(wn, rb + gc) : 0.f;
                pst[3] = vc ? wk[rb + gc] : 0.f;
                pst[4] = vc ? ld_sum(wh, rb + gc) : 0.f;
            }
        }
    };
    auto commit = [&](int buf) {
        unsigned short* dst = Fj + buf * tile_elems;
#pragma unroll
        for (int it = 0; it < NPF; ++it) {
            const int rr = rr0 + it * rstep;
            if (rr < FT) *reinterpret_cast<uint4*>(dst + rr * stride + k0) = pfr[it];
        }
        if (threadIdx.x < FT) {
#pragma unroll
            for (int k = 0; k < 5; ++k) cst[(buf * 5 + k) * FT + threadIdx.x] = pst[k];
        }
    };

    if (nst > 0) { prefetch(0); commit(0); }
    if (nst > 1) prefetch(1);
    __syncthreads();
    for (int s_ = 0; s_ < nst; ++s_) {
        const int buf = s_ & 1, j0 = col0(s_);
        const unsigned short* tile = Fj + buf * tile_elems;
        const unsigned short* At = tile + rl * stride + 8 * half;
        f32x16 acc[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[jt][e] = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const bf16x8_t a = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(At + jt * 32 * stride + 16 * q));
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, bfrag[q]), acc[jt], 0, 0, 0);
            }
        }
        // pair weights, in place: acc[jt][4 g + e] <- W(i, column jt*32 + 8 g + 4 half + e)
        const float* cb = cst + buf * 5 * FT;
        const bool teacher_tile = cross && (s_ & 1);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int jl = jt * 32 + 8 * g + 4 * half;
                const float4 k4 = *reinterpret_cast<const float4*>(cb + FT + jl);
                const float mk4[4] = {k4.x, k4.y, k4.z, k4.w};
                if (teacher_tile) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sx = acc[jt][4 * g + e];
                        const bool hard = vi && j0 + jl + e < N && mk4[e] != mrow && sx > thr;
                        acc[jt][4 * g + e] = hard ? cross_scale * __builtin_amdgcn_rcpf(1.f - sx + 1e-18f) : 0.f;
                    }
                } else {
                    const float4 m4 = *reinterpret_cast<const float4*>(cb + jl);
                    const float4 n4 = *reinterpret_cast<const float4*>(cb + 2 * FT + jl);
                    const float4 q4 = *reinterpret_cast<const float4*>(cb + 3 * FT + jl);
                    const float4 h4 = *reinterpret_cast<const float4*>(cb + 4 * FT + jl);
                    const float mj4[4] = {m4.x, m4.y, m4.z, m4.w}, nj4[4] = {n4.x, n4.y, n4.z, n4.w};
                    const float kj4[4] = {q4.x, q4.y, q4.z, q4.w}, hj4[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // branch-free: both classes' terms are evaluated and one is selected.  The two directions share their
                        // exponentials with the other-class term; a divergent if / else made every lane walk both paths anyway
                        // (82 % of the pairs are same-class at the reference's 10 % foreground) with 3 exps per path and ~4 exec-mask
                        // branch sequences per pair.  All operands are finite for masked-out pairs too (zero rows / columns).
                        const int gj = j0 + jl + e;
                        const float l = acc[jt][4 * g + e] * inv_tau;
                        const float x1 = l - mj4[e], x2 = l - mi;
                        const float e1 = fexp<true>(x1), e2 = fexp<true>(x2);
                        const float r1 = __builtin_amdgcn_rcpf(e1 + nrow), r2 = __builtin_amdgcn_rcpf(e2 + nj4[e]);
                        float same;
                        if (focal) {                             // (uniform)  kappa n r^2 * a dphi,  a dphi = n r (2 a log P - n)
                            const float lg1 = x1 + flog<true>(r1), lg2 = x2 + flog<true>(r2);      // log P = x - log d = x + log r
                            const float t1 = nrow * r1, t2 = nj4[e] * r2;
                            same = krow * t1 * t1 * r1 * (2.f * e1 * lg1 - nrow) + kj4[e] * t2 * t2 * r2 * (2.f * e2 * lg2 - nj4[e]);
                        } else {
                            same = -(krow * nrow * r1 + kj4[e] * nj4[e] * r2);              // dphi = -1 / P
                        }
                        const float diff = e1 * hrow + e2 * hj4[e];
                        const float tv = mk4[e] == mrow ? same : diff;
                        acc[jt][4 * g + e] = (vi && gj < N && gj != gi) ? tv * stud_scale : 0.f;
                    }
                }
            }
        // gradient GEMM: gacc[dt] (M = 32 features, N = the wave's 32 rows) += F_J^T (A, transposed LDS reads) x W^T (B, from acc)
        const unsigned short* Arow = tile + (4 * half + ((lane & 15) >> 2)) * stride + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                unsigned wb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {                 // k-slot 4 g' + e <- register 4 (2 sk + g') + e
                    const int r0 = 8 * sk + 2 * u;
                    wb[u] = (unsigned)f32_to_bf16_bits(acc[jt][r0]) | ((unsigned)f32_to_bf16_bits(acc[jt][r0 + 1]) << 16);
                }
                const bf16x8_t wfrag = __builtin_bit_cast(bf16x8_t, make_uint4(wb[0], wb[1], wb[2], wb[3]));
                const unsigned short* a0 = Arow + (jt * 32 + 16 * sk) * stride;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(a0 + 32 * dt));
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(a0 + 32 * dt + 8 * stride));
                    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    gacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, v), wfrag, gacc[dt], 0, 0, 0);
                }
            }
        if (s_ + 1 < nst) commit(buf ^ 1);
        if (s_ + 2 < nst) prefetch(s_ + 2);
        __syncthreads();
    }
    // gacc[dt][4 g + e] = gf[row gi][feature 32 dt + 8 g + 4 half + e]
    if (vi) {
        float* gb = GS + (((long long)cs * Btot + b) * N + gi) * Dm;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(gb + 32 * dt + 8 * g + 4 * half) =
                    make_float4(gacc[dt][4 * g], gacc[dt][4 * g + 1], gacc[dt][4 * g + 2], gacc[dt][4 * g + 3]);
    }
}

// g_feat = sum over the column-split slabs (ordered)
template <typename T>
__global__ void fecl_combine_kernel(const float* __restrict__ GS, int CS, long long n, T* __restrict__ GF) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = GS[i];
        for (int z = 1; z < CS; ++z) v += GS[z * n + i];
        stf(GF + i, v);
    }
}

__global__ void fecl_finalize_kernel(const double* __restrict__ out, double BN, float lambda_cross, int has_teacher,
                                     float* __restrict__ loss) {
    double l = out[0] / BN;
    if (has_teacher) l += (double)lambda_cross * out[1] / (out[2] + 1e-18);
    loss[0] = (float)l;
}

// =================================================================================================
// C ABI
// =================================================================================================
static inline int lgrid(long long n) {
    long long b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int dycon_seg_losses_fwd(const float* s_logits, const float* t_logits, const void* labels, int label_bytes, int B,
                                    int LB, long long V, float beta, double* sums, int fast, dycon_stream_t stream) {
    DYCON_REQUIRE(s_logits && t_logits && sums && B > 0 && LB >= 0 && LB <= B && V > 0, "seg_losses_fwd: bad arguments");
    DYCON_REQUIRE(LB == 0 || labels, "seg_losses_fwd: labels missing");
    DYCON_REQUIRE(label_bytes == 1 || label_bytes == 8, "seg_losses_fwd: labels must be uint8 or int64");
    if (hipMemsetAsync(sums, 0, 16 * sizeof(double), stream) != hipSuccess) { dycon_set_error("seg_losses_fwd: memset failed"); return DYCON_ERR_LAUNCH; }
    // grid: as many waves in flight as the chip holds -- the kernel is bound by its exp / log sequences (a 512-workgroup grid that
    // thinned the 11 same-address double atomics per workgroup was 1.5x SLOWER)
    const int grid = lgrid((long long)B * V);
    if (fast)
        seg_losses_fwd_kernel<true><<<grid, 256, 0, stream>>>((const float2*)s_logits, (const float2*)t_logits, labels, label_bytes, B, LB, V, beta, sums);
    else
        seg_losses_fwd_kernel<false><<<grid, 256, 0, stream>>>((const float2*)s_logits, (const float2*)t_logits, labels, label_bytes, B, LB, V, beta, sums);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_seg_losses_bwd(const float* s_logits, const float* t_logits, const void* labels, int label_bytes, int B,
                                    int LB, long long V, float beta, const double* sums, const float* coef, int cons_kind,
                                    float* g_logits, int fast, dycon_stream_t stream) {
    DYCON_REQUIRE(s_logits && t_logits && sums && coef && g_logits && B > 0 && LB >= 0 && LB <= B && V > 0, "seg_losses_bwd: bad arguments");
    DYCON_REQUIRE(label_bytes == 1 || label_bytes == 8, "seg_losses_bwd: labels must be uint8 or int64");
    const int grid = lgrid((long long)B * V);
    if (fast)
        seg_losses_bwd_kernel<true><<<grid, 256, 0, stream>>>((const float2*)s_logits, (const float2*)t_logits, labels, label_bytes, B, LB, V, beta, sums, coef, cons_kind, (float2*)g_logits);
    else
        seg_losses_bwd_kernel<false><<<grid, 256, 0, stream>>>((const float2*)s_logits, (const float2*)t_logits, labels, label_bytes, B, LB, V, beta, sums, coef, cons_kind, (float2*)g_logits);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_seg_losses_finalize(const double* sums, int B, int LB, long long V, float beta, float* vals,
                                         dycon_stream_t stream) {
    DYCON_REQUIRE(sums && vals && B > 0 && LB >= 0 && LB <= B && V > 0, "seg_losses_finalize: bad arguments");
    seg_losses_finalize_kernel<<<1, 1, 0, stream>>>(sums, B, LB, V, beta, vals);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_step_loss(const float* vals, const float* fecl, float l_weight, float cons_weight, float u_weight,
                               int dice_kind, int cons_kind, float* out, int* nonfinite, dycon_stream_t stream) {
    DYCON_REQUIRE(vals && out, "step_loss: bad arguments");
    step_loss_kernel<<<1, 1, 0, stream>>>(vals, fecl, l_weight, cons_weight, u_weight, dice_kind, cons_kind, out, nonfinite);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_step_losses(const double* sums, const double* fecl_out, int B, int LB, long long V, float beta, double fecl_rows,
                                 float lambda_cross, int has_teacher, float l_weight, float cons_weight, float u_weight,
                                 int dice_kind, int cons_kind, float* out, int* nonfinite, dycon_stream_t stream) {
    DYCON_REQUIRE(sums && out && B > 0 && LB >= 0 && LB <= B && V > 0 && (!fecl_out || fecl_rows > 0), "step_losses: bad arguments");
    step_losses_kernel<<<1, 1, 0, stream>>>(sums, fecl_out, B, LB, V, beta, fecl_rows, lambda_cross, has_teacher, l_weight, cons_weight,
                                            u_weight, dice_kind, cons_kind, out, nonfinite);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_l2norm_fwd(const void* x, void* y, float* norms, int dtype, long long R, int C, float eps,
                                dycon_stream_t stream) {
    DYCON_REQUIRE(x && y && norms && R > 0 && C > 0, "l2norm_fwd: bad arguments");
    DYCON_DISPATCH(dtype, { l2norm_fwd_kernel<T><<<cdiv(R, 4), 256, 0, stream>>>((const T*)x, (T*)y, norms, R, C, eps); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_l2norm_bwd(const void* y, const float* norms, const void* gy, void* gx, int dtype, long long R, int C,
                                float eps, dycon_stream_t stream) {
    DYCON_REQUIRE(y && norms && gy && gx && R > 0 && C > 0, "l2norm_bwd: bad arguments");
    DYCON_DISPATCH(dtype, { l2norm_bwd_kernel<T><<<cdiv(R, 4), 256, 0, stream>>>((const T*)y, norms, (const T*)gy, (T*)gx, R, C, eps); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_mask_pool(const void* labels, int label_bytes, float* mask, int B, int D, int H, int W, int kd, int kh,
                               int kw, dycon_stream_t stream) {
    DYCON_REQUIRE(labels && mask && B > 0 && kd > 0 && kh > 0 && kw > 0 && D >= kd && H >= kh && W >= kw, "mask_pool: bad arguments");
    DYCON_REQUIRE(label_bytes == 1 || label_bytes == 8, "mask_pool: labels must be uint8 or int64");
    const long long total = (long long)B * (D / kd) * (H / kh) * (W / kw);
    mask_pool_kernel<<<cdiv(total, 4), 256, 0, stream>>>(labels, label_bytes, mask, B, D, H, W, kd, kh, kw);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

// column splits: enough workgroups to occupy the chip twice over (LDS allows 2 per CU).  r128 (the 128-row kernels; their gradient
// pass holds ONE workgroup per CU): few row blocks (N = 1728: 14 per sample) -> as many splits as keep the grid within one round
// of 256, otherwise the split count nearest to 512 workgroups.
static void fecl_split(int B, int N, bool r128, int& CS, int& tps) {
    const int nct = (N + FT - 1) / FT;
    const long long blocks = (long long)((N + (r128 ? 128 : FT) - 1) / (r128 ? 128 : FT)) * B;
    long long want = (512 + blocks - 1) / blocks;
    if (r128) want = blocks <= 128 ? 256 / blocks : (512 + blocks / 2) / blocks;
    if (want > 8) want = 8;
    if (want > nct) want = nct;
    if (want < 1) want = 1;
    tps = (nct + (int)want - 1) / (int)want;
    CS = (nct + tps - 1) / tps;
}

extern "C" size_t dycon_fecl_workspace(int B, int N, int Dm) {
    int CS, CS2, tps;
    fecl_split(B, N, false, CS, tps);
    fecl_split(B, N, true, CS2, tps);
    if (CS2 > CS) CS = CS2;                      // whichever kernel family the dtype / size selects later
    return ((size_t)(4 * CS + 1) * B * N + (size_t)CS * B * N * Dm) * sizeof(float);
}

template <typename T> static size_t fecl_lds_bytes(int Dm, bool grad) {
    typedef FeclTile<T> G;
    const int Dp = (Dm + G::KS - 1) / G::KS * G::KS, stride = Dp + G::PAD;
    size_t n = ((size_t)2 * FT * stride + (grad ? FT * (FT + G::TPAD) : 0)) * sizeof(typename G::E);
    n = (n + 15) / 16 * 16 + 5 * FT * sizeof(float);   // + per-column statistics of the current tile
    return n < 256 ? 256 : n;   // the pass-3 block reduction borrows the first floats
}

template <typename T, int PASS>
static int fecl_launch(const void* feat, const void* teacher, const float* mask, const float* gamb, int B, int N, int Dm,
                       float tau, float gamma, int focal, float thr, float* ws, double* out, const float* coef, float lambda_cross,
                       dycon_stream_t stream) {
    const size_t lds = fecl_lds_bytes<T>(Dm, PASS == 4);
    // > 64 KiB of dynamic LDS needs the opt-in attribute (idempotent, no sync, capture-safe)
    if (hipFuncSetAttribute((const void*)fecl_kernel<T, PASS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        dycon_set_error("fecl: cannot reserve %zu bytes of LDS", lds);
        return DYCON_ERR_LAUNCH;
    }
    int CS, tps;
    fecl_split(B, N, false, CS, tps);
    float* gslab = ws + (size_t)(4 * CS + 1) * B * N;
    dim3 grid(cdiv(N, FT), B, CS);
    fecl_kernel<T, PASS><<<grid, 256, lds, stream>>>((const T*)feat, (const T*)teacher, mask, gamb, N, Dm, tau, gamma, focal, thr, ws,
                                                    out, coef, lambda_cross, gslab, B, CS, tps);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

// passes 1-3 on the 128-row kernel: bf16, a feature dim whose rows split into equal 16-byte columns per thread, the focal weight the
// reference runs (gamma = 2) or none, and N >= 1024 (8 row blocks per sample; the headline N = 1728 runs 0.27 -> 0.25 ms alone
// and the step 4.72 -> 4.64 ms with them, profiles/r03_fecl_n15680.txt)
static bool fecl_rows128_ok(int dtype, int B, int N, int Dm, float gamma, int focal) {
    static const long long min_n = [] { const char* v = getenv("DYCON_FECL_ROWS128_MIN_N"); return v && *v ? atoll(v) : 1024LL; }();
    return dtype == DYCON_BF16 && (Dm == 64 || Dm == 128 || Dm == 256) && (!focal || gamma == 2.f) && N >= min_n;
}
template <int PASS, int NQ>
static int fecl_rows128_launch_nq(const void* feat, const void* teacher, const float* mask, const float* gamb, int B, int N, int Dm,
                                  float tau, int focal, float thr, float* ws, double* out, dycon_stream_t stream) {
    const size_t lds = ((size_t)2 * FT * (Dm + F2_PAD) * 2 + 15) / 16 * 16 + 2 * 2 * FT * sizeof(float);
    if (hipFuncSetAttribute((const void*)fecl_rows128_kernel<PASS, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        dycon_set_error("fecl: cannot reserve %zu bytes of LDS", lds);
        return DYCON_ERR_LAUNCH;
    }
    int CS, tps;
    fecl_split(B, N, true, CS, tps);
    dim3 grid(cdiv(N, F2_ROWS), B, CS);
    fecl_rows128_kernel<PASS, NQ><<<grid, 256, lds, stream>>>((const bf16*)feat, (const bf16*)teacher, mask, gamb, N, Dm, tau, focal, thr, ws,
                                                              out, B, CS, tps);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}
template <int PASS>
static int fecl_rows128_launch(const void* feat, const void* teacher, const float* mask, const float* gamb, int B, int N, int Dm,
                               float tau, int focal, float thr, float* ws, double* out, dycon_stream_t stream) {
    if (Dm == 256) return fecl_rows128_launch_nq<PASS, 16>(feat, teacher, mask, gamb, B, N, Dm, tau, focal, thr, ws, out, stream);
    if (Dm == 128) return fecl_rows128_launch_nq<PASS, 8>(feat, teacher, mask, gamb, B, N, Dm, tau, focal, thr, ws, out, stream);
    return fecl_rows128_launch_nq<PASS, 4>(feat, teacher, mask, gamb, B, N, Dm, tau, focal, thr, ws, out, stream);
}

template <int NQ>
static int fecl_rows128_grad_launch_nq(const void* feat, const void* teacher, const float* mask, int B, int N, int Dm, float tau, int focal,
                                       float thr, float* ws, const double* out, const float* coef, float lambda_cross, dycon_stream_t stream) {
    const size_t lds = ((size_t)2 * FT * (Dm + F2_PAD) * 2 + 15) / 16 * 16 + 2 * 5 * FT * sizeof(float);
    if (hipFuncSetAttribute((const void*)fecl_rows128_grad_kernel<NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        dycon_set_error("fecl: cannot reserve %zu bytes of LDS", lds);
        return DYCON_ERR_LAUNCH;
    }
    int CS, tps;
    fecl_split(B, N, true, CS, tps);
    float* gslab = ws + (size_t)(4 * CS + 1) * B * N;
    dim3 grid(cdiv(N, F2_ROWS), B, CS);
    fecl_rows128_grad_kernel<NQ><<<grid, 256, lds, stream>>>((const bf16*)feat, (const bf16*)teacher, mask, N, Dm, tau, focal, thr, ws, out, coef,
                                                             lambda_cross, gslab, B, CS, tps);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}
static int fecl_rows128_grad_launch(const void* feat, const void* teacher, const float* mask, int B, int N, int Dm, float tau, int focal,
                                    float thr, float* ws, const double* out, const float* coef, float lambda_cross, dycon_stream_t stream) {
    if (Dm == 256) return fecl_rows128_grad_launch_nq<16>(feat, teacher, mask, B, N, Dm, tau, focal, thr, ws, out, coef, lambda_cross, stream);
    if (Dm == 128) return fecl_rows128_grad_launch_nq<8>(feat, teacher, mask, B, N, Dm, tau, focal, thr, ws, out, coef, lambda_cross, stream);
    return fecl_rows128_grad_launch_nq<4>(feat, teacher, mask, B, N, Dm, tau, focal, thr, ws, out, coef, lambda_cross, stream);
}

static int fecl_check(const char* who, const void* feat, const float* mask, int B, int N, int Dm, float tau, size_t ws_bytes) {
    DYCON_REQUIRE(feat && mask && B > 0 && N > 0 && Dm > 0 && tau > 0.f, "%s: bad arguments", who);
    DYCON_REQUIRE(Dm <= 256, "%s: feature dim %d > 256 not supported", who, Dm);
    DYCON_REQUIRE(B <= 65535, "%s: batch too large", who);
    DYCON_REQUIRE(ws_bytes >= dycon_fecl_workspace(B, N, Dm), "%s: workspace too small", who);
    return DYCON_OK;
}

extern "C" int dycon_fecl_fwd(const void* feat, const void* teacher, const float* mask, const float* gambling, int dtype, int B,
                              int N, int Dm, float temperature, float gamma, int use_focal, float cross_thresh,
                              float lambda_cross, double* out, float* loss, float* workspace, size_t ws_bytes,
                              dycon_stream_t stream) {
    if (int e = fecl_check("fecl_fwd", feat, mask, B, N, Dm, temperature, ws_bytes)) return e;
    DYCON_REQUIRE(out && loss && workspace, "fecl_fwd: null pointer");
    const int focal = use_focal && !gambling;   // the gambling branch overrides the focal result (dycon_losses.py:209-211)
    if (hipMemsetAsync(out, 0, 4 * sizeof(double), stream) != hipSuccess) { dycon_set_error("fecl_fwd: memset failed"); return DYCON_ERR_LAUNCH; }
    int e = DYCON_OK;
    if (fecl_rows128_ok(dtype, B, N, Dm, gamma, focal)) {
        e = fecl_rows128_launch<1>(feat, nullptr, mask, gambling, B, N, Dm, temperature, focal, cross_thresh, workspace, out, stream);
        if (!e) e = fecl_rows128_launch<2>(feat, nullptr, mask, gambling, B, N, Dm, temperature, focal, cross_thresh, workspace, out, stream);
        if (!e) e = fecl_rows128_launch<3>(feat, teacher, mask, gambling, B, N, Dm, temperature, focal, cross_thresh, workspace, out, stream);
    } else
    DYCON_DISPATCH(dtype, {
        e = fecl_launch<T, 1>(feat, nullptr, mask, gambling, B, N, Dm, temperature, gamma, focal, cross_thresh, workspace, out, nullptr, lambda_cross, stream);
        if (!e) e = fecl_launch<T, 2>(feat, nullptr, mask, gambling, B, N, Dm, temperature, gamma, focal, cross_thresh, workspace, out, nullptr, lambda_cross, stream);
        if (!e) e = fecl_launch<T, 3>(feat, teacher, mask, gambling, B, N, Dm, temperature, gamma, focal, cross_thresh, workspace, out, nullptr, lambda_cross, stream);
    });
    if (e) return e;
    fecl_finalize_kernel<<<1, 1, 0, stream>>>(out, (double)B * N, lambda_cross, teacher != nullptr, loss);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_fecl_finalize(const double* out, double rows, float lambda_cross, int has_teacher, float* loss,
                                   dycon_stream_t stream) {
    DYCON_REQUIRE(out && loss && rows > 0, "fecl_finalize: bad arguments");
    fecl_finalize_kernel<<<1, 1, 0, stream>>>(out, rows, lambda_cross, has_teacher, loss);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_fecl_bwd(const void* feat, const void* teacher, const float* mask, const float* gambling, int dtype, int B,
                              int N, int Dm, float temperature, float gamma, int use_focal, float cross_thresh,
                              float lambda_cross, const double* out, const float* coef, void* g_feat, float* workspace,
                              size_t ws_bytes, dycon_stream_t stream) {
    if (int e = fecl_check("fecl_bwd", feat, mask, B, N, Dm, temperature, ws_bytes)) return e;
    DYCON_REQUIRE(out && coef && g_feat && workspace, "fecl_bwd: null pointer");
    const int focal = use_focal && !gambling;
    int e = DYCON_OK;
    const bool r128 = fecl_rows128_ok(dtype, B, N, Dm, gamma, focal);
    DYCON_DISPATCH(dtype, {
        if (r128)
            e = fecl_rows128_grad_launch(feat, teacher, mask, B, N, Dm, temperature, focal, cross_thresh, workspace, out, coef, lambda_cross, stream);
        else
            e = fecl_launch<T, 4>(feat, teacher, mask, gambling, B, N, Dm, temperature, gamma, focal, cross_thresh, workspace,
                                  const_cast<double*>(out), coef, lambda_cross, stream);
        if (!e) {
            int CS, tps;
            fecl_split(B, N, r128, CS, tps);
            const long long n = (long long)B * N * Dm;
            fecl_combine_kernel<T><<<lgrid(n), 256, 0, stream>>>(workspace + (size_t)(4 * CS + 1) * B * N, CS, n, (T*)g_feat);
        }
    });
    if (e) return e;
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}
