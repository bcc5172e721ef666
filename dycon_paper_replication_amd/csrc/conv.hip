// Convolution family as implicit GEMM on the gfx950 matrix cores.
//
//   y[row, n] = bias[n] + sum_{t, c} x[src(row, t), c] * W[t, c, n]
//
// rows are voxels (NDHWC activations: a voxel's channels are contiguous, so every MFMA A-fragment
// is one 16-byte load per lane), K = taps*Cin, N = Cout (or 8*Cout for the transposed conv, whose
// epilogue scatters to the doubled grid).  The same kernel therefore serves
//   conv k3 / k2s2 / 1x1 forward, conv-transpose k2s2 forward, and all of their data-gradients
// (VNet.py:16,73,100,175; networks/utils.py:104,107; UNet3D_contrastive.py:249-250,262,265).
// Weights are pre-packed (dycon_pack_bfrag) in MFMA B-fragment order, so a wave reads 1 KiB
// contiguous per fragment straight from L2 -- no LDS round trip for an operand that is shared by
// every workgroup of the launch.
//
// fp32 storage uses v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain: the 1e-4 parity mode),
// bf16 storage uses v_mfma_f32_16x16x32_bf16 (fp32 accumulate).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Frag;
template <> struct Frag<float> { static constexpr int G = 4, KC = 16; };
template <> struct Frag<bf16> { static constexpr int G = 8, KC = 32; };

__device__ __forceinline__ void mma(f32x4& acc, const Vec16<float>& a, const Vec16<float>& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v.x, b.v.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v.y, b.v.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v.z, b.v.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v.w, b.v.w, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x4& acc, const Vec16<bf16>& a, const Vec16<bf16>& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v),
                                                  acc, 0, 0, 0);
}

template <int MODE> __device__ __forceinline__ int n_taps() { return MODE == DYCON_CONV_K3 ? 27 : MODE == DYCON_CONV_K2S2 ? 8 : 1; }

// source voxel of (row voxel (z,y,x), tap t); false when it falls into the zero padding
template <int MODE>
__device__ __forceinline__ bool src_voxel(int t, int z, int y, int x, int Di, int Hi, int Wi, int& zi, int& yi, int& xi) {
    if (MODE == DYCON_CONV_K3) {
        const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
        zi = z + dz - 1; yi = y + dy - 1; xi = x + dx - 1;
        return (unsigned)zi < (unsigned)Di && (unsigned)yi < (unsigned)Hi && (unsigned)xi < (unsigned)Wi;
    } else if (MODE == DYCON_CONV_K2S2) {
        zi = 2 * z + (t >> 2); yi = 2 * y + ((t >> 1) & 1); xi = 2 * x + (t & 1);
        return true;
    } else {
        zi = z; yi = y; xi = x;
        return true;
    }
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_bfrag_kernel(const float* __restrict__ w, T* __restrict__ out, int Tn, int Cin, int N, int N0,
                                  long long s_t, long long s_c, long long s_n1, long long s_n0, int flip, int NT,
                                  long long total) {
    constexpr int G = Frag<T>::G, KC = Frag<T>::KC;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i % G);
        long long q = i / G;
        const int lane = (int)(q % 64);
        q /= 64;
        const int nt = (int)(q % NT);
        const int kc = (int)(q / NT);
        const int k = kc * KC + G * (lane >> 4) + e;
        const int n = nt * 16 + (lane & 15);
        float v = 0.f;
        if (k < Tn * Cin && n < N) {
            int t = k / Cin;
            const int c = k - t * Cin;
            if (flip) t = Tn - 1 - t;
            v = w[t * s_t + c * s_c + (long long)(n / N0) * s_n1 + (long long)(n % N0) * s_n0];
        }
        stf(out + i, v);
    }
}

__global__ void pack_tcn_kernel(const float* __restrict__ w, float* __restrict__ out, int Tn, int Cin, int N, int N0,
                                long long s_t, long long s_c, long long s_n1, long long s_n0, int flip, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const long long q = i / N;
        const int c = (int)(q % Cin);
        int t = (int)(q / Cin);
        if (flip) t = Tn - 1 - t;
        out[i] = w[t * s_t + c * s_c + (long long)(n / N0) * s_n1 + (long long)(n % N0) * s_n0];
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA gather-GEMM
// ------------------------------------------------------------------------------------------------
constexpr int NTB = 4;  // n-tiles (of 16 columns) per workgroup

template <typename T, int MODE, bool SCATTER>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const T* __restrict__ X, const T* __restrict__ Wf,
                                                        const float* __restrict__ bias, T* __restrict__ Y, int B, int Di,
                                                        int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int N, int Cout,
                                                        int NT, int nKC, int accumulate) {
    constexpr int G = Frag<T>::G, KC = Frag<T>::KC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const long long M = (long long)B * Do * Ho * Wo;
    const long long m = (long long)blockIdx.x * 64 + wave * 16 + r;
    const bool mvalid = m < M;
    int xo, yo, zo, bo;
    {
        long long q = mvalid ? m : 0;
        xo = (int)(q % Wo); q /= Wo;
        yo = (int)(q % Ho); q /= Ho;
        zo = (int)(q % Do);
        bo = (int)(q / Do);
    }
    const int nt0 = blockIdx.y * NTB;
    const int Tn = n_taps<MODE>();
    f32x4 acc[NTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int c = G * kg, t = 0;  // this lane's (tap, channel) inside the flattened K = taps*Cin
    while (c >= Cin) { c -= Cin; ++t; }
    const T* wf = Wf + ((long long)nt0 * 64 + lane) * G;
    for (int kc = 0; kc < nKC; ++kc) {
        Vec16<T> a;
        a.v = decltype(a.v){};
        if (mvalid && t < Tn) {
            int zi, yi, xi;
            if (src_voxel<MODE>(t, zo, yo, xo, Di, Hi, Wi, zi, yi, xi))
                a = ld16(X + ((((long long)bo * Di + zi) * Hi + yi) * Wi + xi) * Cin + c);
        }
#pragma unroll
        for (int j = 0; j < NTB; ++j) {
            if (nt0 + j < NT) {
                const Vec16<T> b = ld16(wf + (long long)j * 64 * G);
                mma(acc[j], a, b);
            }
        }
        wf += (long long)NT * 64 * G;
        c += KC;
        while (c >= Cin) { c -= Cin; ++t; }
    }

    // epilogue: C/D layout of the 16x16 tile: column = lane&15, row = 4*(lane>>4) + i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long mo = (long long)blockIdx.x * 64 + wave * 16 + kg * 4 + i;
        if (mo >= M) continue;
        long long rowbase;
        int ox = 0, oy = 0, oz = 0, ob = 0;
        if (SCATTER) {
            long long q = mo;
            ox = (int)(q % Wo); q /= Wo;
            oy = (int)(q % Ho); q /= Ho;
            oz = (int)(q % Do);
            ob = (int)(q / Do);
            rowbase = 0;
        } else {
            rowbase = mo * N;
        }
#pragma unroll
        for (int j = 0; j < NTB; ++j) {
            if (nt0 + j >= NT) continue;
            const int n = (nt0 + j) * 16 + r;
            if (n >= N) continue;
            float v = acc[j][i];
            long long off;
            if (SCATTER) {
                const int tap = n / Cout, co = n - tap * Cout;
                if (bias) v += bias[co];
                const long long ov = (((long long)ob * (2 * Do) + 2 * oz + (tap >> 2)) * (2 * Ho) + 2 * oy + ((tap >> 1) & 1)) * (2 * Wo) +
                                     2 * ox + (tap & 1);
                off = ov * Cout + co;
            } else {
                if (bias) v += bias[n];
                off = rowbase + n;
            }
            if (accumulate) v += ldf(Y + off);
            stf(Y + off, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// skinny direct kernel: one thread per output row, all N (<= 16 per pass) columns
// ------------------------------------------------------------------------------------------------
template <typename TI, typename TO, int MODE>
__global__ __launch_bounds__(256) void conv_direct_kernel(const TI* __restrict__ X, const float* __restrict__ W,
                                                          const float* __restrict__ bias, TO* __restrict__ Y, int B, int Di,
                                                          int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int N,
                                                          int accumulate) {
    const long long M = (long long)B * Do * Ho * Wo;
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    long long q = m;
    const int xo = (int)(q % Wo); q /= Wo;
    const int yo = (int)(q % Ho); q /= Ho;
    const int zo = (int)(q % Do);
    const int bo = (int)(q / Do);
    const int Tn = n_taps<MODE>();
    for (int n0 = 0; n0 < N; n0 += 16) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        const int nn = min(16, N - n0);
        for (int t = 0; t < Tn; ++t) {
            int zi, yi, xi;
            if (!src_voxel<MODE>(t, zo, yo, xo, Di, Hi, Wi, zi, yi, xi)) continue;
            const TI* xp = X + ((((long long)bo * Di + zi) * Hi + yi) * Wi + xi) * Cin;
            const float* wp = W + (long long)t * Cin * N + n0;
            for (int c = 0; c < Cin; ++c) {
                const float xv = ldf(xp + c);
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < nn) acc[j] += xv * wp[(long long)c * N + j];
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j >= nn) continue;
            float v = acc[j] + (bias ? bias[n0 + j] : 0.f);
            TO* yp = Y + m * N + n0 + j;
            if (accumulate) v += ldf(yp);
            stf(yp, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: partial[s][t][ci][co] = sum over the voxel slice s
// MFMA f32 16x16x4 with K = 4 voxels per instruction: A[row=ci][k=voxel], B[k=voxel][col=co]
// -- both operands are read in their natural NDHWC order (16 lanes = 16 consecutive channels).
// ------------------------------------------------------------------------------------------------
template <typename TX, typename TG, int MODE>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const TX* __restrict__ X, const TG* __restrict__ GY,
                                                         float* __restrict__ part, int B, int Di, int Hi, int Wi, int Cin,
                                                         int Do, int Ho, int Wo, int Cout, int nCoBlk, long long rows_per_split) {
    __shared__ float red[4 * NTB * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int ci0 = (blockIdx.x / nCoBlk) * 16;
    const int co0 = (blockIdx.x % nCoBlk) * 16 * NTB;
    const int t = blockIdx.y;
    const int split = blockIdx.z;
    const long long M = (long long)B * Do * Ho * Wo;
    const long long m_beg = split * rows_per_split;
    const long long m_end = min(M, m_beg + rows_per_split);
    const int Tn = n_taps<MODE>();
    f32x4 acc[NTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntile = min(NTB, (Cout - co0 + 15) / 16);

    for (long long mq = m_beg + 4 * wave; mq < m_end; mq += 16) {
        const long long m = mq + kg;
        float a = 0.f;
        float b[NTB];
#pragma unroll
        for (int j = 0; j < NTB; ++j) b[j] = 0.f;
        if (m < m_end) {
            long long q = m;
            const int xo = (int)(q % Wo); q /= Wo;
            const int yo = (int)(q % Ho); q /= Ho;
            const int zo = (int)(q % Do);
            const int bo = (int)(q / Do);
            int zi, yi, xi;
            if (src_voxel<MODE>(t, zo, yo, xo, Di, Hi, Wi, zi, yi, xi))
                a = ldf(X + ((((long long)bo * Di + zi) * Hi + yi) * Wi + xi) * Cin + ci0 + r);
            const TG* gp = GY + m * Cout + co0 + r;
#pragma unroll
            for (int j = 0; j < NTB; ++j)
                if (j < ntile) b[j] = ldf(gp + 16 * j);
        }
#pragma unroll
        for (int j = 0; j < NTB; ++j)
            if (j < ntile) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], acc[j], 0, 0, 0);
    }
    // cross-wave reduction through LDS; tile element (row=ci = 4*kg+i, col=co = r)
#pragma unroll
    for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(wave * NTB + j) * 256 + (kg * 4 + i) * 16 + r] = acc[j][i];
    __syncthreads();
    float* dst = part + (((long long)split * Tn + t) * Cin) * Cout;
    for (int e = threadIdx.x; e < ntile * 256; e += 256) {
        const int j = e >> 8, ci = (e >> 4) & 15, co = e & 15;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(w * NTB + j) * 256 + ci * 16 + co];
        dst[(long long)(ci0 + ci) * Cout + co0 + 16 * j + co] = s;
    }
}

// skinny weight gradient: thread j owns output element (t, ci, co); the block walks a voxel slice
template <typename TX, typename TG, int MODE>
__global__ __launch_bounds__(256) void conv_wgrad_direct_kernel(const TX* __restrict__ X, const TG* __restrict__ GY,
                                                                float* __restrict__ part, int B, int Di, int Hi, int Wi,
                                                                int Cin, int Do, int Ho, int Wo, int Cout,
                                                                long long rows_per_split) {
    const int Tn = n_taps<MODE>();
    const int L = Tn * Cin * Cout;
    const int j = blockIdx.y * blockDim.x + threadIdx.x;
    const long long M = (long long)B * Do * Ho * Wo;
    const long long m_beg = (long long)blockIdx.x * rows_per_split;
    const long long m_end = min(M, m_beg + rows_per_split);
    if (j >= L) return;
    const int co = j % Cout, ci = (j / Cout) % Cin, t = j / (Cout * Cin);
    float acc = 0.f;
    long long q = m_beg;
    int xo = (int)(q % Wo); q /= Wo;
    int yo = (int)(q % Ho); q /= Ho;
    int zo = (int)(q % Do);
    int bo = (int)(q / Do);
    for (long long m = m_beg; m < m_end; ++m) {
        int zi, yi, xi;
        if (src_voxel<MODE>(t, zo, yo, xo, Di, Hi, Wi, zi, yi, xi))
            acc += ldf(X + ((((long long)bo * Di + zi) * Hi + yi) * Wi + xi) * Cin + ci) * ldf(GY + m * Cout + co);
        if (++xo == Wo) { xo = 0; if (++yo == Ho) { yo = 0; if (++zo == Do) { zo = 0; ++bo; } } }
    }
    part[(long long)blockIdx.x * L + j] = acc;
}

// out[map(i)] = sum_p part[p*L + i], i = (t*Cin + ci)*Cout + co, map = t*s_t + ci*s_c + co*s_n
__global__ void reduce_partials_kernel(const float* __restrict__ part, int P, int L, float* __restrict__ out, int Cin,
                                       int Cout, long long s_t, long long s_c, long long s_n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += part[(long long)p * L + i];
    const int co = i % Cout, ci = (i / Cout) % Cin, t = i / (Cout * Cin);
    out[t * s_t + ci * s_c + co * s_n] = s;
}

// column sums of an (rows, C) matrix, stage 1: partial[blk][c]
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, float* __restrict__ part, long long rows, int C,
                                                     long long rows_per_block) {
    __shared__ float sm[256];
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    if (C >= 256) {
        for (int c = threadIdx.x; c < C; c += 256) {
            float s = 0.f;
            for (long long r = r0; r < r1; ++r) s += ldf(X + r * C + c);
            part[(long long)blockIdx.x * C + c] = s;
        }
        return;
    }
    const int rpi = 256 / C;  // rows per iteration
    const int rr = threadIdx.x / C, c = threadIdx.x % C;
    float s = 0.f;
    if (rr < rpi)
        for (long long r = r0 + rr; r < r1; r += rpi) s += ldf(X + r * C + c);
    sm[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
        float tot = 0.f;
        for (int k = 0; k < rpi; ++k) tot += sm[k * C + threadIdx.x];
        part[(long long)blockIdx.x * C + threadIdx.x] = tot;
    }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static void row_grid(int mode, int Di, int Hi, int Wi, int& Do, int& Ho, int& Wo) {
    if (mode == DYCON_CONV_K2S2) { Do = Di / 2; Ho = Hi / 2; Wo = Wi / 2; }
    else { Do = Di; Ho = Hi; Wo = Wi; }
}

extern "C" size_t dycon_bfrag_bytes(int dtype, int T, int Cin, int N) {
    const int KC = dtype == DYCON_BF16 ? 32 : 16, es = dtype == DYCON_BF16 ? 2 : 4;
    const long long nKC = ((long long)T * Cin + KC - 1) / KC, NT = (N + 15) / 16;
    return (size_t)(nKC * NT * 64 * (16 / es) * es);
}

extern "C" int dycon_pack_bfrag(const float* w, void* out, int dtype, int Tn, int Cin, int N, int N0, long long s_t,
                                long long s_c, long long s_n1, long long s_n0, int flip_taps, dycon_stream_t stream) {
    DYCON_REQUIRE(w && out && Tn > 0 && Cin > 0 && N > 0 && N0 > 0, "pack_bfrag: bad arguments");
    DYCON_DISPATCH(dtype, {
        constexpr int G = Frag<T>::G, KC = Frag<T>::KC;
        const int NT = (N + 15) / 16;
        const long long nKC = ((long long)Tn * Cin + KC - 1) / KC;
        const long long total = nKC * NT * 64 * G;
        const int grid = cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256);
        pack_bfrag_kernel<T><<<grid, 256, 0, stream>>>(w, (T*)out, Tn, Cin, N, N0, s_t, s_c, s_n1, s_n0, flip_taps, NT, total);
    });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_pack_tcn(const float* w, float* out, int T, int Cin, int N, int N0, long long s_t, long long s_c,
                              long long s_n1, long long s_n0, int flip_taps, dycon_stream_t stream) {
    DYCON_REQUIRE(w && out && T > 0 && Cin > 0 && N > 0 && N0 > 0, "pack_tcn: bad arguments");
    const long long total = (long long)T * Cin * N;
    pack_tcn_kernel<<<cdiv(total, 256), 256, 0, stream>>>(w, out, T, Cin, N, N0, s_t, s_c, s_n1, s_n0, flip_taps, total);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

template <typename T, int MODE, bool SC>
static void launch_gemm(const void* x, const void* wf, const float* bias, void* y, int accumulate, int B, int Di, int Hi,
                        int Wi, int Cin, int N, int Cout, dycon_stream_t stream) {
    int Do, Ho, Wo;
    row_grid(MODE, Di, Hi, Wi, Do, Ho, Wo);
    const long long M = (long long)B * Do * Ho * Wo;
    const int NT = (N + 15) / 16;
    const int Tn = MODE == DYCON_CONV_K3 ? 27 : MODE == DYCON_CONV_K2S2 ? 8 : 1;
    const int nKC = (Tn * Cin + Frag<T>::KC - 1) / Frag<T>::KC;
    dim3 grid(cdiv(M, 64), cdiv(NT, NTB));
    conv_gemm_kernel<T, MODE, SC><<<grid, 256, 0, stream>>>((const T*)x, (const T*)wf, bias, (T*)y, B, Di, Hi, Wi, Cin, Do, Ho,
                                                            Wo, N, Cout, NT, nKC, accumulate);
}

extern "C" int dycon_conv_gemm(const void* x, const void* wfrag, const float* bias, void* y, int dtype, int mode,
                               int scatter, int accumulate, int B, int Di, int Hi, int Wi, int Cin, int N, int Cout,
                               dycon_stream_t stream) {
    DYCON_REQUIRE(x && wfrag && y, "conv_gemm: null pointer");
    DYCON_REQUIRE(B > 0 && Di > 0 && Hi > 0 && Wi > 0 && Cin > 0 && N > 0 && Cout > 0, "conv_gemm: bad shape");
    DYCON_REQUIRE(mode >= 0 && mode <= 2, "conv_gemm: bad mode %d", mode);
    DYCON_REQUIRE(Cin % (dtype == DYCON_BF16 ? 8 : 4) == 0, "conv_gemm: Cin=%d not a multiple of the fragment width", Cin);
    DYCON_REQUIRE(N % 16 == 0, "conv_gemm: N=%d not a multiple of 16 (use dycon_conv_direct)", N);
    DYCON_REQUIRE(!scatter || (mode == DYCON_CONV_1X1 && N == 8 * Cout), "conv_gemm: scatter needs mode 1x1 and N == 8*Cout");
    DYCON_REQUIRE(scatter || N == Cout, "conv_gemm: N must equal Cout without scatter");
    DYCON_REQUIRE(mode != DYCON_CONV_K2S2 || (Di % 2 == 0 && Hi % 2 == 0 && Wi % 2 == 0), "conv_gemm: k2s2 needs even dims");
    DYCON_DISPATCH(dtype, {
        if (scatter) launch_gemm<T, DYCON_CONV_1X1, true>(x, wfrag, bias, y, accumulate, B, Di, Hi, Wi, Cin, N, Cout, stream);
        else if (mode == DYCON_CONV_K3) launch_gemm<T, DYCON_CONV_K3, false>(x, wfrag, bias, y, accumulate, B, Di, Hi, Wi, Cin, N, Cout, stream);
        else if (mode == DYCON_CONV_K2S2) launch_gemm<T, DYCON_CONV_K2S2, false>(x, wfrag, bias, y, accumulate, B, Di, Hi, Wi, Cin, N, Cout, stream);
        else launch_gemm<T, DYCON_CONV_1X1, false>(x, wfrag, bias, y, accumulate, B, Di, Hi, Wi, Cin, N, Cout, stream);
    });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

template <typename TI, typename TO>
static void launch_direct(const void* x, const float* w, const float* bias, void* y, int mode, int accumulate, int B, int Di,
                          int Hi, int Wi, int Cin, int N, dycon_stream_t stream) {
    int Do, Ho, Wo;
    row_grid(mode, Di, Hi, Wi, Do, Ho, Wo);
    const long long M = (long long)B * Do * Ho * Wo;
    const int grid = cdiv(M, 256);
#define DYCON_DIRECT(MODE) \
    conv_direct_kernel<TI, TO, MODE><<<grid, 256, 0, stream>>>((const TI*)x, w, bias, (TO*)y, B, Di, Hi, Wi, Cin, Do, Ho, Wo, N, accumulate)
    if (mode == DYCON_CONV_K3) DYCON_DIRECT(DYCON_CONV_K3);
    else if (mode == DYCON_CONV_K2S2) DYCON_DIRECT(DYCON_CONV_K2S2);
    else DYCON_DIRECT(DYCON_CONV_1X1);
#undef DYCON_DIRECT
}

extern "C" int dycon_conv_direct(const void* x, int x_dtype, const float* w_tcn, const float* bias, void* y, int y_dtype,
                                 int mode, int accumulate, int B, int Di, int Hi, int Wi, int Cin, int N,
                                 dycon_stream_t stream) {
    DYCON_REQUIRE(x && w_tcn && y, "conv_direct: null pointer");
    DYCON_REQUIRE(B > 0 && Di > 0 && Hi > 0 && Wi > 0 && Cin > 0 && N > 0, "conv_direct: bad shape");
    DYCON_REQUIRE(mode >= 0 && mode <= 2, "conv_direct: bad mode %d", mode);
    if (x_dtype == DYCON_F32 && y_dtype == DYCON_F32) launch_direct<float, float>(x, w_tcn, bias, y, mode, accumulate, B, Di, Hi, Wi, Cin, N, stream);
    else if (x_dtype == DYCON_BF16 && y_dtype == DYCON_F32) launch_direct<bf16, float>(x, w_tcn, bias, y, mode, accumulate, B, Di, Hi, Wi, Cin, N, stream);
    else if (x_dtype == DYCON_F32 && y_dtype == DYCON_BF16) launch_direct<float, bf16>(x, w_tcn, bias, y, mode, accumulate, B, Di, Hi, Wi, Cin, N, stream);
    else if (x_dtype == DYCON_BF16 && y_dtype == DYCON_BF16) launch_direct<bf16, bf16>(x, w_tcn, bias, y, mode, accumulate, B, Di, Hi, Wi, Cin, N, stream);
    else { dycon_set_error("conv_direct: bad dtypes %d %d", x_dtype, y_dtype); return DYCON_ERR_INVALID; }
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

// ---- weight gradient planning (shared by the workspace query and the launch)
struct WgradPlan { bool mfma; int splits; long long rows_per_split; int L; };
static WgradPlan wgrad_plan(int mode, int B, int Di, int Hi, int Wi, int Cin, int Cout) {
    int Do, Ho, Wo;
    row_grid(mode, Di, Hi, Wi, Do, Ho, Wo);
    const long long M = (long long)B * Do * Ho * Wo;
    const int Tn = mode == DYCON_CONV_K3 ? 27 : mode == DYCON_CONV_K2S2 ? 8 : 1;
    WgradPlan p;
    p.L = Tn * Cin * Cout;
    p.mfma = (Cin % 16 == 0) && (Cout % 16 == 0);
    if (p.mfma) {
        const long long blocks_xy = (long long)(Cin / 16) * cdiv(Cout, 16 * NTB) * Tn;
        long long want = (2048 + blocks_xy - 1) / blocks_xy;           // aim for ~2k workgroups
        long long max_by_rows = (M + 255) / 256;                       // >= 256 rows per split
        long long s = want < 1 ? 1 : want;
        if (s > max_by_rows) s = max_by_rows;
        if (s > 256) s = 256;
        if (s < 1) s = 1;
        long long rps = (M + s - 1) / s;
        rps = (rps + 15) / 16 * 16;
        p.splits = (int)((M + rps - 1) / rps);
        p.rows_per_split = rps;
    } else {
        long long rps = 4096;
        p.splits = (int)((M + rps - 1) / rps);
        p.rows_per_split = rps;
    }
    return p;
}

extern "C" size_t dycon_conv_wgrad_workspace(int mode, int B, int Di, int Hi, int Wi, int Cin, int Cout) {
    const WgradPlan p = wgrad_plan(mode, B, Di, Hi, Wi, Cin, Cout);
    return (size_t)p.splits * p.L * sizeof(float);
}

template <typename TX, typename TG>
static void launch_wgrad(const void* x, const void* gy, float* part, int mode, const WgradPlan& p, int B, int Di, int Hi,
                         int Wi, int Cin, int Cout, dycon_stream_t stream) {
    int Do, Ho, Wo;
    row_grid(mode, Di, Hi, Wi, Do, Ho, Wo);
    const int Tn = mode == DYCON_CONV_K3 ? 27 : mode == DYCON_CONV_K2S2 ? 8 : 1;
    if (p.mfma) {
        const int nCoBlk = cdiv(Cout, 16 * NTB);
        dim3 grid((Cin / 16) * nCoBlk, Tn, p.splits);
#define DYCON_WG(MODE) \
    conv_wgrad_kernel<TX, TG, MODE><<<grid, 256, 0, stream>>>((const TX*)x, (const TG*)gy, part, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, nCoBlk, p.rows_per_split)
        if (mode == DYCON_CONV_K3) DYCON_WG(DYCON_CONV_K3);
        else if (mode == DYCON_CONV_K2S2) DYCON_WG(DYCON_CONV_K2S2);
        else DYCON_WG(DYCON_CONV_1X1);
#undef DYCON_WG
    } else {
        dim3 grid(p.splits, cdiv(p.L, 256));
#define DYCON_WGD(MODE) \
    conv_wgrad_direct_kernel<TX, TG, MODE><<<grid, 256, 0, stream>>>((const TX*)x, (const TG*)gy, part, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, p.rows_per_split)
        if (mode == DYCON_CONV_K3) DYCON_WGD(DYCON_CONV_K3);
        else if (mode == DYCON_CONV_K2S2) DYCON_WGD(DYCON_CONV_K2S2);
        else DYCON_WGD(DYCON_CONV_1X1);
#undef DYCON_WGD
    }
}

extern "C" int dycon_conv_wgrad(const void* x, int x_dtype, const void* gy, int g_dtype, float* dw, int mode, int B, int Di,
                                int Hi, int Wi, int Cin, int Cout, long long s_t, long long s_c, long long s_n,
                                float* workspace, size_t ws_bytes, dycon_stream_t stream) {
    DYCON_REQUIRE(x && gy && dw && workspace, "conv_wgrad: null pointer");
    DYCON_REQUIRE(B > 0 && Di > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "conv_wgrad: bad shape");
    DYCON_REQUIRE(mode >= 0 && mode <= 2, "conv_wgrad: bad mode %d", mode);
    const WgradPlan p = wgrad_plan(mode, B, Di, Hi, Wi, Cin, Cout);
    DYCON_REQUIRE(ws_bytes >= (size_t)p.splits * p.L * sizeof(float), "conv_wgrad: workspace too small (%zu < %zu)", ws_bytes,
                  (size_t)p.splits * p.L * sizeof(float));
    if (x_dtype == DYCON_F32 && g_dtype == DYCON_F32) launch_wgrad<float, float>(x, gy, workspace, mode, p, B, Di, Hi, Wi, Cin, Cout, stream);
    else if (x_dtype == DYCON_BF16 && g_dtype == DYCON_BF16) launch_wgrad<bf16, bf16>(x, gy, workspace, mode, p, B, Di, Hi, Wi, Cin, Cout, stream);
    else if (x_dtype == DYCON_BF16 && g_dtype == DYCON_F32) launch_wgrad<bf16, float>(x, gy, workspace, mode, p, B, Di, Hi, Wi, Cin, Cout, stream);
    else if (x_dtype == DYCON_F32 && g_dtype == DYCON_BF16) launch_wgrad<float, bf16>(x, gy, workspace, mode, p, B, Di, Hi, Wi, Cin, Cout, stream);
    else { dycon_set_error("conv_wgrad: bad dtypes"); return DYCON_ERR_INVALID; }
    DYCON_LAUNCH_CHECK();
    reduce_partials_kernel<<<cdiv(p.L, 256), 256, 0, stream>>>(workspace, p.splits, p.L, dw, Cin, Cout, s_t, s_c, s_n);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

static void colsum_plan(long long rows, int C, int& blocks, long long& rpb) {
    long long b = (rows + 511) / 512;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    rpb = (rows + b - 1) / b;
    blocks = (int)((rows + rpb - 1) / rpb);
}

extern "C" size_t dycon_colsum_workspace(long long rows, int C) {
    int blocks; long long rpb;
    colsum_plan(rows, C, blocks, rpb);
    return (size_t)blocks * C * sizeof(float);
}

extern "C" int dycon_colsum(const void* x, int dtype, float* out, long long rows, int C, float* workspace, size_t ws_bytes,
                            dycon_stream_t stream) {
    DYCON_REQUIRE(x && out && workspace && rows > 0 && C > 0, "colsum: bad arguments");
    int blocks; long long rpb;
    colsum_plan(rows, C, blocks, rpb);
    DYCON_REQUIRE(ws_bytes >= (size_t)blocks * C * sizeof(float), "colsum: workspace too small");
    DYCON_DISPATCH(dtype, { colsum_kernel<T><<<blocks, 256, 0, stream>>>((const T*)x, workspace, rows, C, rpb); });
    DYCON_LAUNCH_CHECK();
    reduce_partials_kernel<<<cdiv(C, 256), 256, 0, stream>>>(workspace, blocks, C, out, 1, C, 0, 0, 1);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}
