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
#include <stdlib.h>

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

// all weight packs of a step in ONE launch: jobs[] lives in device memory (built once; pointers are stable arena views).
// One thread per FRAGMENT PIECE (the G = 8 / 4 consecutive k of one lane): the (k-chunk, n-tile, lane) decode and the column's
// source offset are computed once per piece and (tap, channel) advance incrementally -- with one thread per element the kernel was
// bound by ~10 integer divisions per value (52 us for 57 MB at the head of every step's dependent chain).
__global__ __launch_bounds__(256) void pack_batch_kernel(const dycon_pack_job_t* __restrict__ jobs) {
    const dycon_pack_job_t jb = jobs[blockIdx.y];
    if (jb.kind == 2) {          // plain fp32 [T][Cin][N]
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < jb.total; i += (long long)gridDim.x * 256) {
            const int n = (int)(i % jb.N);
            const long long q = i / jb.N;
            const int c = (int)(q % jb.Cin);
            int t = (int)(q / jb.Cin);
            if (jb.flip) t = jb.T - 1 - t;
            ((float*)jb.out)[i] = jb.w[t * jb.s_t + c * jb.s_c + (long long)(n / jb.N0) * jb.s_n1 + (long long)(n % jb.N0) * jb.s_n0];
        }
        return;
    }
    const int G = jb.kind == 0 ? 8 : 4, KC = jb.kind == 0 ? 32 : 16;
    const long long pieces = jb.total / G;
    const int K = jb.T * jb.Cin;
    for (long long pi = (long long)blockIdx.x * 256 + threadIdx.x; pi < pieces; pi += (long long)gridDim.x * 256) {
        const int lane = (int)(pi & 63);
        const long long q = pi >> 6;
        const int nt = (int)(q % jb.NT);
        const int kc = (int)(q / jb.NT);
        const int n = nt * 16 + (lane & 15);
        int k = kc * KC + G * (lane >> 4);
        int t = k / jb.Cin, c = k - t * jb.Cin;
        const bool ncol = n < jb.N;
        const long long noff = ncol ? (long long)(n / jb.N0) * jb.s_n1 + (long long)(n % jb.N0) * jb.s_n0 : 0;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = 0.f;
            if (e < G && ncol && k < K) {
                const int tt = jb.flip ? jb.T - 1 - t : t;
                v[e] = jb.w[tt * jb.s_t + c * jb.s_c + noff];
            }
            ++k;
            if (++c == jb.Cin) { c = 0; ++t; }
        }
        if (jb.kind == 0) {
            uint4 o;
            o.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
            o.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
            o.z = (unsigned)f32_to_bf16_bits(v[4]) | ((unsigned)f32_to_bf16_bits(v[5]) << 16);
            o.w = (unsigned)f32_to_bf16_bits(v[6]) | ((unsigned)f32_to_bf16_bits(v[7]) << 16);
            reinterpret_cast<uint4*>(jb.out)[pi] = o;
        } else {
            reinterpret_cast<float4*>(jb.out)[pi] = make_float4(v[0], v[1], v[2], v[3]);
        }
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
                                                        int NT, int nKC, int accumulate, float* __restrict__ slab,
                                                        int kc_per_split, int span_major = 0) {
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

    // split-K (small spatial levels): blockIdx.z owns k-chunks [kc0, kc1) and writes an fp32 partial slab
    const int kc0 = slab ? blockIdx.z * kc_per_split : 0;
    const int kc1 = slab ? min(nKC, kc0 + kc_per_split) : nKC;
    const int kl = kc0 * KC + G * kg;          // this lane's position inside the flattened K = taps*Cin
    int t = kl / Cin, c = kl - t * Cin;
    const T* wf = Wf + (((long long)kc0 * NT + nt0) * 64 + lane) * G;
    // k-chunks in batches of 4: the A gathers and B fragments of a batch are all requested before its first MFMA
    const int tstep = KC / Cin, cstep = KC - tstep * Cin;     // per-chunk advance of (tap, channel)
    constexpr int UB = 2;
    for (int kcb = kc0; kcb < kc1; kcb += UB) {
        Vec16<T> a[UB], bq[UB][NTB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            a[u].v = decltype(a[u].v){};
            if (kcb + u < kc1) {
                if (mvalid && t < Tn) {
                    int zi, yi, xi;
                    if (src_voxel<MODE>(t, zo, yo, xo, Di, Hi, Wi, zi, yi, xi))
                        a[u] = ld16(X + ((((long long)bo * Di + zi) * Hi + yi) * Wi + xi) * Cin + c);
                }
#pragma unroll
                for (int j = 0; j < NTB; ++j)
                    if (nt0 + j < NT) bq[u][j] = ld16(wf + (long long)j * 64 * G);
                wf += (long long)NT * 64 * G;
                t += tstep;
                c += cstep;
                if (c >= Cin) { c -= Cin; ++t; }
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (kcb + u < kc1) {
#pragma unroll
                for (int j = 0; j < NTB; ++j)
                    if (nt0 + j < NT) mma(acc[j], a[u], bq[u][j]);
            }
        }
    }

    // epilogue: C/D layout of the 16x16 tile: column = lane&15, row = 4*(lane>>4) + i
    if (slab) {
        float* sl = slab + (long long)blockIdx.z * M * N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long mo = (long long)blockIdx.x * 64 + wave * 16 + kg * 4 + i;
            if (mo >= M) continue;
#pragma unroll
            for (int j = 0; j < NTB; ++j) {
                const int n = (nt0 + j) * 16 + r;
                // span_major (finish deferred to the one-launch norm): [8-channel span][row][8] -- that norm's workgroups own 8-16
                // channels of one sample and then read their share of every slab as ONE contiguous run
                if (nt0 + j < NT && n < N) sl[span_major ? ((long long)(n >> 3) * M + mo) * 8 + (n & 7) : mo * N + n] = acc[j][i];
            }
        }
        return;
    }
    // Stage the 64 x (NTB*16) tile through LDS (one channel per lane in the D fragments) and write 16-byte channel
    // groups: a lane-per-element epilogue is store-issue bound (2-byte stores) on the big levels.
    constexpr int VN = Vec16<T>::N;
    constexpr int OS = NTB * 16 + VN;                       // padded LDS row stride (elements)
    __shared__ __attribute__((aligned(16))) T Ot[64 * OS];
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
        const int n = (nt0 + j) * 16 + r;
        float bv = 0.f;
        if (bias && nt0 + j < NT && n < N) bv = bias[SCATTER ? n % Cout : n];
#pragma unroll
        for (int i = 0; i < 4; ++i) stf(Ot + (wave * 16 + kg * 4 + i) * OS + j * 16 + r, acc[j][i] + bv);
    }
    __syncthreads();
    constexpr int PPR = NTB * 16 / VN;                      // 16-byte pieces per row
    for (int e = threadIdx.x; e < 64 * PPR; e += 256) {
        const int row = e / PPR, pc = e % PPR;
        const long long mo = (long long)blockIdx.x * 64 + row;
        const int n0 = nt0 * 16 + pc * VN;
        if (mo >= M || n0 >= N) continue;
        long long off;
        if (SCATTER) {
            long long q = mo;
            const int ox = (int)(q % Wo); q /= Wo;
            const int oy = (int)(q % Ho); q /= Ho;
            const int oz = (int)(q % Do);
            const int ob = (int)(q / Do);
            const int tap = n0 / Cout, co = n0 - tap * Cout;   // Cout % VN == 0: a piece never straddles two taps
            const long long ov = (((long long)ob * (2 * Do) + 2 * oz + (tap >> 2)) * (2 * Ho) + 2 * oy + ((tap >> 1) & 1)) * (2 * Wo) +
                                 2 * ox + (tap & 1);
            off = ov * Cout + co;
        } else {
            off = mo * N + n0;
        }
        Vec16<T> o = ld16(Ot + row * OS + pc * VN);
        if (accumulate) {
            const Vec16<T> old = ld16(Y + off);
#pragma unroll
            for (int k = 0; k < VN; ++k) o.set(k, o.get(k) + old.get(k));
        }
        st16(Y + off, o);
    }
}

// ------------------------------------------------------------------------------------------------
// k=3 convolution at the small spatial levels (12^3, 6^3: 128 / 256 channels), bf16: LDS-tiled split-K GEMM.
//
// There the implicit GEMM is weight-dominated (M = B*V = 6912 or 864 rows, K = 27*Cin = 3456 .. 6912, N = 128 / 256): in the
// gather kernel above every WAVE streams its own copy of the B fragments from L2 (13 FLOP per L2 byte).  Here a workgroup
// owns a 64-row x 128-column tile; per 32-wide k-step the A tile (64 gathered 64-byte voxel segments, zero padding resolved
// at staging time) and the B tile (8 packed fragments, 8 KB) are staged ONCE into double-buffered LDS and shared by the four
// waves (2 x 2, each 32 rows x 64 columns = 8 MFMAs per k-step on 6 ds_read_b128): 43 FLOP per L2 byte, one barrier per k-step,
// the next k-step's global loads in flight under the MFMAs.  Split-K partial slabs + ordered finish as in conv_gemm_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int CT_BM = 64, CT_BN = 128, CT_AS = 40;          // A rows padded to 80 B: conflict-free ds_read_b128 fragments
constexpr int CT_OS = CT_BN + 4;                            // fp32 row stride of the slab epilogue's LDS image (528 B: the four row
                                                            // groups of a D fragment land 16 banks apart)

// KS2 = 32-channel chunks per k-step (1 or 2).  With one chunk per step a wave issues 8 MFMAs (128 matrix cycles) between two barriers
// and pays, per step, a publish, a request and ~70 scalar instructions of address bookkeeping: PMC (profiles/r03_pmc_small_levels.txt)
// has 1935 SALU + 1158 VALU instructions per wave against 216 MFMAs, the matrix pipe busy 8.5 % of a wave's life and 42 % of it spent
// in s_waitcnt / s_barrier.  KS2 = 2 (Cin % 64 == 0: the V-Net's 128- and 256-channel levels) halves the number of steps: 16 MFMAs
// per barrier, two chunks of the same tap requested with one address computation (LDS 52 KB per workgroup: 3 per CU).
#ifndef CT_DIAG
#define CT_DIAG 0     // timing diagnostics only (wrong results): 1 no A loads, 2 no B loads, 4 no slab stores, 8 no MFMAs (bit mask)
#endif
template <int KS2>
__global__ __launch_bounds__(256, KS2 == 1 ? 4 : 3) void conv_k3_tile_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wf,
                                                           const float* __restrict__ bias, bf16* __restrict__ Y,
                                                           float* __restrict__ slab, int B, int D, int H, int W, int Cin, int N,
                                                           int NT, int nKC, int kc_per_split, int accumulate, int span_major = 0) {
    // one LDS block: the double-buffered A / B tiles of the k-loop, re-used by the slab epilogue as a 64 x 128 fp32 image
    constexpr int A_ELEMS = CT_BM * CT_AS, B_ELEMS = 8 * 64 * 8;
    constexpr int KLOOP_BYTES = 2 * KS2 * (A_ELEMS + B_ELEMS) * 2, EPI_BYTES = CT_BM * CT_OS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[KLOOP_BYTES > EPI_BYTES ? KLOOP_BYTES : EPI_BYTES];
    unsigned short (*As)[KS2][A_ELEMS] = reinterpret_cast<unsigned short (*)[KS2][A_ELEMS]>(lds_raw);
    unsigned short (*Bs)[KS2][B_ELEMS] = reinterpret_cast<unsigned short (*)[KS2][B_ELEMS]>(lds_raw + 2 * KS2 * A_ELEMS * 2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const long long M = (long long)B * D * H * W;
    const int nt0 = blockIdx.y * 8;
    // staging role: A piece (row, 16-byte quarter of the 32-channel segment), two B pieces
    const int arow = threadIdx.x >> 2, aq = threadIdx.x & 3;
    const long long am = (long long)blockIdx.x * CT_BM + arow;
    int ax, ay, az, ab;
    {
        long long q = am < M ? am : 0;
        ax = (int)(q % W); q /= W;
        ay = (int)(q % H); q /= H;
        az = (int)(q % D);
        ab = (int)(q / D);
    }
    const bool avalid = am < M;
    const int kc0 = slab ? blockIdx.z * kc_per_split : 0;
    const int kc1 = slab ? min(nKC, kc0 + kc_per_split) : nKC;
    int t = (kc0 * 32) / Cin, c = kc0 * 32 - t * Cin;          // (tap, first channel) of the next k-step to request: Cin % (32 KS2) == 0
    int kreq = kc0;                                            // k-steps are requested strictly in order
    // Address generation is kept OFF the vector ALU (measured: 75 VALU + 50 SALU instructions per k-step against 8 MFMAs when the
    // tap is decoded per thread): a k-step's tap offset and weight-chunk base are wave-uniform (scalar registers), the thread's own
    // voxel / fragment-lane offsets are fixed 32-bit values, and the zero padding of all 27 taps is one precomputed bit mask.
    unsigned vmask = 0;
    for (int tt = 0; tt < 27; ++tt) {
        const int zi = az + tt / 9 - 1, yi = ay + (tt / 3) % 3 - 1, xi = ax + tt % 3 - 1;
        if (avalid && (unsigned)zi < (unsigned)D && (unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)W) vmask |= 1u << tt;
    }
    const unsigned aoff = (unsigned)(((((long long)ab * D + az) * H + ay) * W + ax) * Cin + 8 * aq) * 2u;   // bytes; small levels: < 2^31
    const unsigned boff = threadIdx.x * 16u;
    // Two register stages keep the global loads of two k-steps in flight under the MFMAs of a third (more stages spill at the
    // register budget that lets several workgroups share a CU, which hides the rest of the L2 latency).
    struct Stage { uint4 a[KS2], b0[KS2], b1[KS2]; };
    auto request = [&](Stage& st) {
        const int tt = min(t, 26);
        const int dz = tt / 9, dy = (tt / 3) % 3, dx = tt % 3;                    // wave-uniform
        const char* xs = reinterpret_cast<const char*>(X) + (((long long)((dz - 1) * H + (dy - 1)) * W + (dx - 1)) * Cin + c) * 2;
        const char* ws = reinterpret_cast<const char*>(Wf) + ((long long)min(kreq, nKC - KS2) * NT + nt0) * 1024;
        const bool in = ((vmask >> tt) & 1u) != 0;
#pragma unroll
        for (int u = 0; u < KS2; ++u) {                          // the chunks of a step share the tap: channels c + 32 u
            st.a[u] = make_uint4(0, 0, 0, 0);
            if (!(CT_DIAG & 1) && in && kreq + u < kc1) st.a[u] = *reinterpret_cast<const uint4*>(xs + aoff + 64 * u);
            if (!(CT_DIAG & 2) || kreq == kc0) {
                st.b0[u] = *reinterpret_cast<const uint4*>(ws + (long long)u * NT * 1024 + boff);
                st.b1[u] = *reinterpret_cast<const uint4*>(ws + (long long)u * NT * 1024 + boff + 4096);
            }
        }
        kreq += KS2;
        c += 32 * KS2;
        if (c == Cin) { c = 0; ++t; }
    };
    auto publish = [&](const Stage& st, int buf) {
#pragma unroll
        for (int u = 0; u < KS2; ++u) {
            *reinterpret_cast<uint4*>(&As[buf][u][arow * CT_AS + 8 * aq]) = st.a[u];
            *reinterpret_cast<uint4*>(&Bs[buf][u][threadIdx.x * 8]) = st.b0[u];
            *reinterpret_cast<uint4*>(&Bs[buf][u][(threadIdx.x + 256) * 8]) = st.b1[u];
        }
    };
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
#pragma unroll
        for (int u = 0; u < KS2; ++u) {
            bf16x8 a[2], b[4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
                a[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&As[buf][u][(wm * 32 + m * 16 + r) * CT_AS + 8 * kg]));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Bs[buf][u][((wn * 4 + j) * 64 + lane) * 8]));
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (CT_DIAG & 8) acc[m][j][0] += (float)a[m][0] * (float)b[j][0];
                    else acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[j], acc[m][j], 0, 0, 0);
                }
        }
    };

    Stage s0, s1;
    request(s0);
    publish(s0, 0);                                            // k-step 0
    request(s1); request(s0);                                  // k-steps 1, 2 in flight (stage of k-step j: j % 2)
    __syncthreads();
    const int nk = (kc1 - kc0 + KS2 - 1) / KS2;
    // iteration j: MFMAs on LDS buffer j % 2; publish k-step j + 1 into the other buffer; request k-step j + 3 into the freed stage
    for (int j = 0; j < nk; j += 2) {
        compute(0); publish(s1, 1); request(s1); __syncthreads();
        if (j + 1 >= nk) break;
        compute(1); publish(s0, 0); request(s0); __syncthreads();
    }
    // epilogue.  D layout of a 16x16 tile: column = lane & 15, row = 4 * (lane >> 4) + i
    if (slab && span_major == 0) {
        // Partial slab [row][N] fp32: a lane-per-element epilogue writes 64-byte pieces of four rows per store instruction (measured:
        // 7 of the 27 us of a 128 -> 128 @ 12^3 launch + finish, profiles/r03_conv_tile_bounds.txt).  The tile goes through LDS
        // (the k-loop's buffers are free: the loop ends on a barrier) and leaves as 16-byte pieces, two whole 512-byte rows per wave
        // instruction -- with N = 128 the workgroup's 64 rows are one contiguous 32 KB run of the slab.
        float* Ot = reinterpret_cast<float*>(lds_raw);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) Ot[(wm * 32 + m * 16 + kg * 4 + i) * CT_OS + (wn * 4 + j) * 16 + r] = acc[m][j][i];
        __syncthreads();
        float* sl = slab + (long long)blockIdx.z * M * N + (long long)blockIdx.x * CT_BM * N + nt0 * 16;
        const long long rows = min((long long)CT_BM, M - (long long)blockIdx.x * CT_BM);
#pragma unroll
        for (int e = threadIdx.x; e < CT_BM * (CT_BN / 4); e += 256) {
            const int row = e / (CT_BN / 4), pc = e % (CT_BN / 4);
            if (row < rows && !(CT_DIAG & 4))
                *reinterpret_cast<float4*>(sl + (long long)row * N + pc * 4) = *reinterpret_cast<const float4*>(Ot + row * CT_OS + pc * 4);
        }
        return;
    }
    if (slab) {                                                // span_major 1: span-major slabs (finish deferred to the one-launch norm); 2: the
                                                               // lane-per-element row-major epilogue (DYCON_TILE_LDS_EPI=0, kept for A/B timing)
        float* sl = slab + (long long)blockIdx.z * M * N;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long mo = (long long)blockIdx.x * CT_BM + wm * 32 + m * 16 + kg * 4 + i;
                if (mo >= M) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = (nt0 + wn * 4 + j) * 16 + r;
                    if (n < N) sl[span_major == 1 ? ((long long)(n >> 3) * M + mo) * 8 + (n & 7) : mo * N + n] = acc[m][j][i];
                }
            }
        return;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long mo = (long long)blockIdx.x * CT_BM + wm * 32 + m * 16 + kg * 4 + i;
            if (mo >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = (nt0 + wn * 4 + j) * 16 + r;
                if (n >= N) continue;
                float v = acc[m][j][i] + (bias ? bias[n] : 0.f);
                if (accumulate) v += ldf(Y + mo * N + n);
                stf(Y + mo * N + n, v);
            }
        }
}

// ------------------------------------------------------------------------------------------------
// k=3 convolution with an LDS-staged halo tile (bf16): the large spatial levels.
//
// One workgroup = a 4x8x8 block of output voxels (halo 6x10x10) x up to 64 output channels.  The input
// channels are walked in chunks of CK (16 or 32): the chunk's halo tile is staged once (16-byte loads, zero
// padding resolved at staging time), then all 27 taps read their A fragments from LDS with one
// ds_read_b128 per 16x16 tile -- every input byte comes from HBM/L2 once per tile instead of 27 times.
// B fragments (pre-packed, shared by every workgroup) stream from L1/L2.
//   CK = 32: voxel rows padded to 80 B so the 8 x-neighbours of a fragment fall in 8 distinct 16-B bank
//            slots; CK = 16: 32-B rows (8 voxels = one 256-B bank row).
//   wave layout WM x WN over (16 m-tiles) x (NTW n-tiles): 4x1 for Cout <= 32, 2x2 for 64-wide blocks.
// ------------------------------------------------------------------------------------------------
constexpr int CL_TZ = 4, CL_TY = 8, CL_TX = 8;
constexpr int CL_HZ = CL_TZ + 2, CL_HY = CL_TY + 2, CL_HX = CL_TX + 2;
constexpr int CL_NH = CL_HZ * CL_HY * CL_HX;     // 600 halo voxels
constexpr int CL_NV = CL_TZ * CL_TY * CL_TX;     // 256 voxels = 16 m-tiles

#ifndef CL_LB2
#define CL_LB2 2                                   // min workgroups per CU the register allocation must allow: every variant fits 2 (<32,4,2>: 248 VGPRs, no scratch)
#endif
#ifndef CL_NTB64
#define CL_NTB64 4                                 // n-tiles per workgroup for 64-wide output blocks (experiments: 2)
#endif
#ifndef CL_BPREF
#define CL_BPREF 0                                 // B fragments read one k-step ahead of their MFMAs
#endif
template <int CK, int NTB, int WM, int NW = 4>     // NW waves per workgroup: 4, or 8 (two per SIMD) where the grid gives a CU one workgroup
__global__ __launch_bounds__(64 * NW, NW == 4 ? CL_LB2 : 1) void conv_k3_lds_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wf,
                                                          const float* __restrict__ bias, bf16* __restrict__ Y, int B, int D, int H,
                                                          int W, int Cin, int Cout, int NT, int tilesZ, int tilesY, int tilesX,
                                                          int accumulate) {
    constexpr int NTHR = 64 * NW;
    constexpr int WN = NW / WM;                // waves along N
    constexpr int MTW = 16 / WM;               // m-tiles per wave
    constexpr int NTW = NTB / WN;              // n-tiles per wave
    // LDS image of the halo tile, laid out so that every ds_read_b128 lane group (16 lanes = 8 x-neighbours of two y-rows,
    // two channel chunks) hits 16 distinct 16-byte bank slots for every tap (SQ_LDS_BANK_CONFLICT = 0):
    //   CK = 32: 64 B per voxel, x-row pitch 10 voxels (= 8 slots mod 16), the four 16-B chunks of a voxel rotated by its
    //            halo x:  chunk c of voxel hx sits at position (c + hx) & 3
    //   CK = 16: 32 B per voxel, x-row pitch padded to 16 voxels (= 0 slots mod 16), chunks in place
    constexpr int VS = CK;                      // LDS voxel stride in elements
    constexpr int RP = CK == 32 ? CL_HX : 16;   // x-row pitch in voxels
    constexpr int HALO_ELEMS = CL_HZ * CL_HY * RP * VS;
    constexpr int LDS_ELEMS = HALO_ELEMS > CL_NV * (NTB * 16 + 8) ? HALO_ELEMS : CL_NV * (NTB * 16 + 8);
    __shared__ __attribute__((aligned(16))) unsigned short Xh[LDS_ELEMS + (CK == 32 ? 9 : 7) * NTB * 64 * 8];
    unsigned short* Bs = Xh + LDS_ELEMS;       // one group of B fragments: [k-step][n-tile][lane][8]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    // Workgroups are dealt to the 8 XCDs round-robin: give every XCD one contiguous range of tiles, so that the halo overlap
    // between neighbouring tiles is served by that XCD's L2 instead of being fetched into two of them
    int tile = blockIdx.x;
    {
        const int nT = gridDim.x, per = nT >> 3;
        if (per > 0 && tile < per * 8) tile = (tile & 7) * per + (tile >> 3);
    }
    const int tx = tile % tilesX; tile /= tilesX;
    const int ty = tile % tilesY; tile /= tilesY;
    const int tz = tile % tilesZ;
    const int b = tile / tilesZ;
    const int z0 = tz * CL_TZ, y0 = ty * CL_TY, x0 = tx * CL_TX;
    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LDS element offset of this lane's voxel (tap (0,0,0) corner of the halo) for each of its m-tiles
    int vbase[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int v = (wm * MTW + m) * 16 + r;                 // voxel index inside the tile: (z, y, x) = (v/64, (v/8)%8, v%8)
        vbase[m] = (((v >> 6) * CL_HY + ((v >> 3) & 7)) * RP + (v & 7)) * VS;
    }
    const int nChunks = Cin / CK;
    constexpr int NKS = CK == 32 ? 27 : 14;                    // MFMA k-steps (32 wide) per chunk
    constexpr int GRP = CK == 32 ? 9 : 7;                      // k-steps per B group
    constexpr int NG = NKS / GRP;                              // groups per chunk (3 or 2)
    constexpr int PPV = CK / 8;                                // 16-byte pieces per halo voxel
    constexpr int NST = (CL_NH * PPV + NTHR - 1) / NTHR;             // halo pieces per thread
    constexpr int BPIECES = GRP * NTB * 64;                    // 16-byte pieces of one B group (all n-tiles of the workgroup)
    constexpr int NSB = (BPIECES + NTHR - 1) / NTHR;                 // B pieces per thread

    // Register-staged pipeline (issue early / write late): while the MFMAs of phase p run, the global loads of phase p+1
    // (next B group, and the next chunk's halo at a chunk boundary) are already in flight in registers; they are written to
    // LDS after the barrier that retires phase p.  B fragments go through LDS so the four waves fetch them from L2 ONCE.
    uint4 stgH[NST], stgB[NSB];
    auto load_halo = [&](int ch) {
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int e = threadIdx.x + NTHR * it;
            const int hv = e / PPV, pc = e % PPV;
            const int hx = hv % CL_HX, hy = (hv / CL_HX) % CL_HY, hz = hv / (CL_HX * CL_HY);
            const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
            stgH[it] = make_uint4(0, 0, 0, 0);
            const bool in = e < CL_NH * PPV && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            if (in) {
                stgH[it] = *reinterpret_cast<const uint4*>(X + ((((long long)b * D + z) * H + y) * W + x) * Cin + ch * CK + 8 * pc);
            }
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int e = threadIdx.x + NTHR * it;
            if (e < CL_NH * PPV) {
                const int hv = e / PPV, pc = e % PPV;
                const int hx = hv % CL_HX, hyz = hv / CL_HX;
                const int pos = CK == 32 ? ((pc + hx) & 3) : pc;
                *reinterpret_cast<uint4*>(Xh + (hyz * RP + hx) * VS + 8 * pos) = stgH[it];
            }
        }
    };
    auto load_b = [&](int ch, int g) {
#pragma unroll
        for (int it = 0; it < NSB; ++it) {
            const int e = threadIdx.x + NTHR * it;          // piece (u, j, lane)
            const int ln = e & 63, j = (e >> 6) % NTB, u = e / (64 * NTB);
            const int ks = g * GRP + u;
            // CK = 32: flattened K = tap*Cin + channel in 32-chunks.  CK = 16: weights packed chunk-major (each 16-channel
            // chunk as its own 27-tap x 16 problem, 14 k-steps), so 48-channel inputs (U-Net decoder) run here too
            const int kc = CK == 32 ? ks * nChunks + ch : ch * NKS + ks;
            stgB[it] = make_uint4(0, 0, 0, 0);
            if (e < BPIECES)
                stgB[it] = *reinterpret_cast<const uint4*>(Wf + (((long long)kc * NT + blockIdx.y * NTB + j) * 64 + ln) * 8);
        }
    };
    auto store_b = [&]() {
#pragma unroll
        for (int it = 0; it < NSB; ++it) {
            const int e = threadIdx.x + NTHR * it;
            if (e < BPIECES) *reinterpret_cast<uint4*>(Bs + e * 8) = stgB[it];
        }
    };
    auto tap_off = [&](int ks) {
        int t, coff;
        if (CK == 32) { t = ks; coff = 8 * ((kg + (r & 7) + t % 3) & 3); }             // rotated chunk of halo voxel x + dx
        else { t = 2 * ks + (kg >> 1); coff = 8 * (kg & 1); if (t > 26) t = 26; }   // tap 27 is padding: its packed weights are 0
        return ((t / 9) * CL_HY * RP + ((t / 3) % 3) * RP + (t % 3)) * VS + coff;
    };

    load_halo(0);
    load_b(0, 0);
    store_halo();
    store_b();
    __syncthreads();
    const int nPhases = nChunks * NG;
#pragma unroll 1
    for (int ph = 0; ph < nPhases; ++ph) {
        const int ch = ph / NG, g = ph - ch * NG;
        const bool has_next = ph + 1 < nPhases;
        const bool next_chunk = has_next && g == NG - 1;
        if (has_next) load_b(next_chunk ? ch + 1 : ch, next_chunk ? 0 : g + 1);
        if (next_chunk) load_halo(ch + 1);
        // ---- phase p: GRP k-steps from LDS.  A fragments are read one k-step ahead of their MFMAs.
        bf16x8 afr[2][MTW];
        {
            const int toff = tap_off(g * GRP);
#pragma unroll
            for (int m = 0; m < MTW; ++m) afr[0][m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Xh + vbase[m] + toff));
        }
#if CL_BPREF
        bf16x8 bfr[2][NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
            bfr[0][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + ((wn * NTW + j) * 64 + lane) * 8));
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            if (u + 1 < GRP) {
                const int toff = tap_off(g * GRP + u + 1);
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    bfr[(u + 1) & 1][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + (((u + 1) * NTB + wn * NTW + j) * 64 + lane) * 8));
#pragma unroll
                for (int m = 0; m < MTW; ++m)
                    afr[(u + 1) & 1][m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Xh + vbase[m] + toff));
            }
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[u & 1][m], bfr[u & 1][j], acc[m][j], 0, 0, 0);
        }
#else
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            bf16x8 bfr[NTW];
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + ((u * NTB + wn * NTW + j) * 64 + lane) * 8));
            if (u + 1 < GRP) {
                const int toff = tap_off(g * GRP + u + 1);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
                    afr[(u + 1) & 1][m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Xh + vbase[m] + toff));
            }
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[u & 1][m], bfr[j], acc[m][j], 0, 0, 0);
        }
#endif
        if (has_next) {
            __syncthreads();                 // phase p fully consumed
            store_b();
            if (next_chunk) store_halo();
            __syncthreads();
        }
    }
    // ---- epilogue through LDS: the D fragments hold one channel per lane (2-byte scattered stores would be
    //      store-issue bound); stage the 256 x CBW bf16 tile in LDS and write whole 16-byte channel groups
    constexpr int CBW = NTB * 16;                  // channels of this workgroup
    constexpr int OS = CBW + 8;                    // LDS row stride (elements): +16 B against bank conflicts
    __syncthreads();                               // all waves are done reading the halo
    unsigned short* Ot = Xh;
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int nl = (wn * NTW + j) * 16 + r;
            const float bv = bias ? bias[blockIdx.y * CBW + nl] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) Ot[((wm * MTW + m) * 16 + 4 * kg + i) * OS + nl] = f32_to_bf16_bits(acc[m][j][i] + bv);
        }
    __syncthreads();
    constexpr int PPR = CBW / 8;                   // 16-byte pieces per voxel row
#pragma unroll
    for (int it = 0; it < CL_NV * PPR / NTHR; ++it) {
        const int e = threadIdx.x + NTHR * it;
        const int v = e / PPR, pc = e % PPR;
        const int z = z0 + (v >> 6), y = y0 + ((v >> 3) & 7), x = x0 + (v & 7);
        if (z >= D || y >= H || x >= W) continue;
        bf16* yp = Y + ((((long long)b * D + z) * H + y) * W + x) * Cout + blockIdx.y * CBW + 8 * pc;
        Vec16<bf16> o;
        o.v = *reinterpret_cast<const uint4*>(Ot + v * OS + 8 * pc);
        if (accumulate) {
            const Vec16<bf16> old = ld16(yp);
#pragma unroll
            for (int k = 0; k < 8; ++k) o.set(k, o.get(k) + old.get(k));
        }
        st16(yp, o);
    }
}

// ------------------------------------------------------------------------------------------------
// The 96^3 level (16 channels wide, where most of the step's bytes live): persistent, weight-stationary k=3 kernels.
//
// conv_k3_lds_kernel above pays, per 256-voxel tile, an un-overlapped HBM round trip for the halo, a restage of the whole
// weight tensor through LDS and a B-fragment LDS read per MFMA.  At 16 channels the complete weight tensor is 14 B fragments
// = 56 VGPRs, so here
//   * every wave keeps ALL weights in registers for the lifetime of the workgroup (no B traffic at all after the prologue),
//   * a workgroup walks many tiles (grid = 2 workgroups per CU) and the next tile's halo is already in flight in registers
//     while the matrix cores work on the current one (register-staged: written to LDS behind the barrier that retires it),
//   * the output tile leaves through its own LDS staging area, so its 16-byte global stores overlap the next tile's MFMAs,
//   * each XCD owns one contiguous range of tiles: the 2.3x halo overlap between neighbouring tiles is served by that XCD's L2.
// LDS A-fragment reads (one ds_read_b128 per MFMA, the array's limit) and the HBM stream are then the two bounds.
// ------------------------------------------------------------------------------------------------
// Statistics epilogue of the persistent 96^3 kernels (conv_k3_p16 / conv_k3_c1 with STATS): per output channel {sum, sum of squares}
// of the bf16 values the workgroup STORES, accumulated in two registers per lane and n-tile over all tiles of one sample, then
// reduced over the lane groups and the four waves and written as ONE row [channel][2] per (sample, workgroup) of stat_part
// ([B][gridDim.x][CB][2] floats) -- the layout dycon_norm_fwd_parts / dycon_norm_stats_parts finalize.  The normalisation that follows
// then skips its statistics pass over the tensor (at 96^3: 113 MB re-read, ~20 us per site).
template <int NT>
struct TileStats {
    float s1[NT], s2[NT];
    int b;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;
    }
    __device__ __forceinline__ void add(int j, unsigned short bits, bool ok) {
        const float f = ok ? bf16_bits_to_f32(bits) : 0.f;
        s1[j] += f;
        s2[j] += f * f;
    }
    // uniform call (all 256 threads): red = NT * 4 waves * 32 floats of LDS
    __device__ __forceinline__ void flush(float* __restrict__ stat_part, float* red) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float a = s1[j], q = s2[j];
            a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);
            a += __shfl_xor(a, 32, 64); q += __shfl_xor(q, 32, 64);
            if (lane < 16) { red[(wave * NT * 16 + j * 16 + r) * 2] = a; red[(wave * NT * 16 + j * 16 + r) * 2 + 1] = q; }
        }
        __syncthreads();
        if ((int)threadIdx.x < NT * 32) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[w * NT * 32 + threadIdx.x];
            stat_part[((long long)b * gridDim.x + blockIdx.x) * (NT * 32) + threadIdx.x] = v;
        }
        __syncthreads();
        clear();
    }
};

constexpr int P16_RP = 16;                                  // x-row pitch of the halo image in voxels (conflict-free, see above)
constexpr int P16_XH = CL_HZ * CL_HY * P16_RP * 16;         // halo image, bf16 elements (30 KB)
#ifndef P16_WGS
#define P16_WGS 2                                            // workgroups per CU (register budget 256 VGPRs)
#endif
#ifndef P16_DEPTH
#define P16_DEPTH 2                                          // halo tiles in flight per workgroup
#endif

// tiles of this workgroup: its XCD's contiguous range (workgroups are dealt to XCDs round-robin), walked with stride gridDim.x/8
__device__ __forceinline__ void xcd_tile_range(int nTiles, int& first, int& end, int& stride) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per = (nTiles + 7) >> 3;
    stride = gridDim.x >> 3;
    first = xcd * per + slot;
    end = min(nTiles, (xcd + 1) * per);
}

// 16-byte global store the compiler's s_waitcnt bookkeeping does not see.  On gfx9-family targets loads and stores share vmcnt and
// may retire out of order with respect to each other, so with a store pending the compiler waits vmcnt(0) before the first use of
// ANY earlier load -- which would drain the halo prefetches that are meant to stay in flight across tiles.  With the stores hidden
// only loads are tracked (they retire in order) and the compiler emits counted vmcnt(N); the hardware counter additionally holds
// these stores, which can only make a counted wait longer, never shorter.  The store reads its data VGPRs when it issues.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_untracked(void* p, const uint4& v) {
    const u32x4 q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(q) : "memory");
}

struct TileGeo {            // wave-uniform (SGPR) description of one tile
    long long org;          // element offset of voxel (b, z0, y0, x0), 16 channels per voxel
    int z0, y0, x0;
    int inner;              // the whole 6x10x10 halo lies inside the volume: no clamping, no zero padding
};

template <int DEPTH, bool ACC, bool STATS = false>
__global__ __launch_bounds__(256, P16_WGS) void conv_k3_p16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wf,
                                                                   const float* __restrict__ bias, bf16* __restrict__ Y, int B,
                                                                   int D, int H, int W, int tilesZ, int tilesY, int tilesX,
                                                                   int nTiles, float* __restrict__ stat_part = nullptr) {
    constexpr int OS = 16 + 8;                               // output staging row stride (elements): +16 B against conflicts
    __shared__ __attribute__((aligned(16))) unsigned short Xh[P16_XH];
    __shared__ __attribute__((aligned(16))) unsigned short Ot[CL_NV * OS];
    __shared__ float sred[STATS ? 4 * 32 : 1];
    int tile, t_end, t_stride;
    xcd_tile_range(nTiles, tile, t_end, t_stride);
    if (STATS && threadIdx.x < 32)                           // rows of the samples this workgroup never touches: zero
        for (int n = 0; n < B; ++n) stat_part[((long long)n * gridDim.x + blockIdx.x) * 32 + threadIdx.x] = 0.f;
    if (tile >= t_end) return;                               // (uniform) nothing to do: keeps every wait below on a single path
    TileStats<1> ts;
    ts.clear();
    ts.b = -1;
    const int tiles_per_sample = tilesZ * tilesY * tilesX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    bf16x8 bfr[14];                                          // K = 27 taps x 16 channels = 14 k-steps of 32 (tap 27 = zero padding)
#pragma unroll
    for (int ks = 0; ks < 14; ++ks)
        bfr[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Wf + ((long long)ks * 64 + lane) * 8));
    const float bv = bias ? bias[r] : 0.f;
    // A-fragment addresses: wave w owns z-slice w of the tile; its m-tile m is the y-row pair 2m, 2m+1 (16 voxels), so the four
    // m-tiles differ by a constant (an immediate offset of the ds_read) and one address register per k-step suffices.
    // This lane's tap of k-step ks is 2 ks + (kg >> 1); tap 27 is padding (zero weights, reads tap 26).
    auto tap_elems = [](int t) { return ((t / 9) * CL_HY * P16_RP + ((t / 3) % 3) * P16_RP + (t % 3)) * 16; };
    constexpr int MSTEP = 2 * P16_RP * 16;                   // elements between consecutive m-tiles
    int abase[14];
    {
        const int vb = ((wave * CL_HY + (r >> 3)) * P16_RP + (r & 7)) * 16 + 8 * (kg & 1);
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            const int t0 = 2 * ks, t1 = 2 * ks + 1 > 26 ? 26 : 2 * ks + 1;
            abase[ks] = vb + ((kg >> 1) ? tap_elems(t1) : tap_elems(t0));
        }
    }

    // Tile-independent geometry of this thread's NST halo pieces and 2 output pieces, computed ONCE: with 56 MFMAs per wave and
    // tile, per-tile index arithmetic (divisions by the tile counts, piece -> voxel decoding, 64-bit address products) would
    // otherwise cost several times the matrix work in VALU/SALU issue slots (measured: 310 VALU + 218 SALU per tile).
    // Pieces past the 1200th (threads >= 176 of the fifth round) duplicate piece 1199: same address, same LDS slot, same value.
    constexpr int NST = (CL_NH * 2 + 255) / 256;             // 16-byte halo pieces per thread (5)
    int rel[NST], lofs[NST], hco[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int e = min((int)threadIdx.x + 256 * it, CL_NH * 2 - 1);
        const int hv = e >> 1, pc = e & 1;
        const int hx = hv % CL_HX, hy = (hv / CL_HX) % CL_HY, hz = hv / (CL_HX * CL_HY);
        rel[it] = (((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * 16 + 8 * pc;      // element offset from the tile's origin voxel
        lofs[it] = ((hv / CL_HX) * P16_RP + hx) * 16 + 8 * pc;                   // element offset in the LDS image
        hco[it] = hz | (hy << 8) | (hx << 16) | (pc << 24);
    }
    int orel[2], oofs[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = threadIdx.x + 256 * it;
        const int v = e >> 1, pc = e & 1;
        orel[it] = (((v >> 6) * H + ((v >> 3) & 7)) * W + (v & 7)) * 16 + 8 * pc;
        oofs[it] = v * OS + 8 * pc;
    }
    auto geometry = [&](int t, TileGeo& g) {                 // scalar: tile index -> origin / flags (clamped past the end)
        t = min(t, t_end - 1);
        const int tx = t % tilesX; t /= tilesX;
        const int ty = t % tilesY; t /= tilesY;
        const int tz = t % tilesZ;
        const int b = t / tilesZ;
        g.z0 = tz * CL_TZ; g.y0 = ty * CL_TY; g.x0 = tx * CL_TX;
        g.org = ((((long long)b * D + g.z0) * H + g.y0) * W + g.x0) * 16;
        g.inner = g.z0 >= 1 && g.z0 + CL_TZ < D && g.y0 >= 1 && g.y0 + CL_TY < H && g.x0 >= 1 && g.x0 + CL_TX < W;
    };
    // Halo loads are UNCONDITIONAL (boundary tiles read clamped, valid addresses and zero the padding when writing LDS): every wave
    // issues exactly NST loads per tile, so the compiler's counted s_waitcnt vmcnt(N) keeps the younger tiles' loads in flight.
    auto load_halo = [&](const TileGeo& g, uint4 (&stg)[NST]) {
        const bf16* base = X + g.org;
        if (g.inner) {
#pragma unroll
            for (int it = 0; it < NST; ++it) stg[it] = *reinterpret_cast<const uint4*>(base + rel[it]);
        } else {
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int hz = hco[it] & 255, hy = (hco[it] >> 8) & 255, hx = (hco[it] >> 16) & 255, pc = hco[it] >> 24;
                const int z = min(max(g.z0 + hz - 1, 0), D - 1), y = min(max(g.y0 + hy - 1, 0), H - 1), x = min(max(g.x0 + hx - 1, 0), W - 1);
                const int dz = z - g.z0, dy = y - g.y0, dx = x - g.x0;
                stg[it] = *reinterpret_cast<const uint4*>(base + ((dz * H + dy) * W + dx) * 16 + 8 * pc);
            }
        }
    };
    auto store_halo = [&](const TileGeo& g, const uint4 (&stg)[NST]) {
        if (g.inner) {
#pragma unroll
            for (int it = 0; it < NST; ++it) *reinterpret_cast<uint4*>(Xh + lofs[it]) = stg[it];
        } else {
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int hz = hco[it] & 255, hy = (hco[it] >> 8) & 255, hx = (hco[it] >> 16) & 255;
                const int z = g.z0 + hz - 1, y = g.y0 + hy - 1, x = g.x0 + hx - 1;
                const bool in = (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                uint4 v = stg[it];
                if (!in) v = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(Xh + lofs[it]) = v;
            }
        }
    };

    // One tile: [issue the halo loads of tile + DEPTH*stride into set `ld`] -> MFMAs from the LDS image -> barrier -> stage the
    // output, write the halo of tile + stride (set `st`, requested DEPTH - 1 tiles ago) -> barrier -> 16-byte global stores.
    auto one_tile = [&](int cur, TileGeo& gcur, TileGeo& gld, uint4 (&ld)[NST], const TileGeo& gst, const uint4 (&st)[NST]) {
        geometry(cur + DEPTH * t_stride, gld);
        load_halo(gld, ld);
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            bf16x8 afr[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) afr[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Xh + abase[ks] + m * MSTEP));
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[m], bfr[ks], acc[m], 0, 0, 0);
        }
        __syncthreads();                                     // halo image consumed; previous tile's Ot fully stored
        const bool full = gcur.z0 + CL_TZ <= D && gcur.y0 + CL_TY <= H && gcur.x0 + CL_TX <= W;
        if (STATS && cur < t_end) {                          // (uniform) a new sample begins: publish the finished one's sums
            const int bs = cur / tiles_per_sample;
            if (bs != ts.b) { if (ts.b >= 0) ts.flush(stat_part, sred); ts.b = bs; }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned short bits = f32_to_bf16_bits(acc[m][i] + bv);
                Ot[((wave * 4 + m) * 16 + 4 * kg + i) * OS + r] = bits;
                if (STATS) {                                 // voxel (z = wave, y = 2 m + (4 kg + i) / 8, x = (4 kg + i) % 8) of the tile
                    const int vy = 2 * m + ((4 * kg + i) >> 3), vx = (4 * kg + i) & 7;
                    ts.add(0, bits, cur < t_end && (full || (gcur.z0 + wave < D && gcur.y0 + vy < H && gcur.x0 + vx < W)));
                }
            }
        store_halo(gst, st);
        __syncthreads();
        bf16* ybase = Y + gcur.org;
        if (cur < t_end) {                                   // (else: idle slot of the unrolled tail -- a clamped tile, nothing stored)
#pragma unroll
            for (int it = 0; it < 2; ++it) {                 // 256 voxels x 2 pieces of 16 B
                if (!full) {
                    const int v = (threadIdx.x + 256 * it) >> 1;
                    if (gcur.z0 + (v >> 6) >= D || gcur.y0 + ((v >> 3) & 7) >= H || gcur.x0 + (v & 7) >= W) continue;
                }
                bf16* yp = ybase + orel[it];
                Vec16<bf16> o;
                o.v = *reinterpret_cast<const uint4*>(Ot + oofs[it]);
                if (ACC) {                                   // (drains the prefetches: the gradient-accumulate variant is not pipelined)
                    const Vec16<bf16> old = ld16(yp);
#pragma unroll
                    for (int k = 0; k < 8; ++k) o.set(k, o.get(k) + old.get(k));
                    st16(yp, o);
                } else {
                    st16_untracked(yp, o.v);
                }
            }
        }
        gcur = gst;                                          // the image now in LDS
    };

    // register sets: tile j of this workgroup's sequence travels in set j % DEPTH (with its scalar geometry)
    uint4 s0[NST], s1[NST], s2[NST];
    TileGeo g0, g1, g2, gcur;
    geometry(tile, g0);
    load_halo(g0, s0);
    store_halo(g0, s0);                                      // (the wait for s0 also retires the weight loads)
    gcur = g0;
    if (DEPTH >= 2) { geometry(tile + t_stride, g1); load_halo(g1, s1); }
    if (DEPTH == 3) { geometry(tile + 2 * t_stride, g2); load_halo(g2, s2); }
    __syncthreads();
    if (DEPTH == 1) {
        for (; tile < t_end; tile += t_stride) one_tile(tile, gcur, g0, s0, g0, s0);   // load j+1 at the top, write it at the end
    } else if (DEPTH == 2) {
        for (; tile < t_end; tile += 2 * t_stride) {         // tile j: loads j+2 into set j%2 (free since j was written), writes j+1
            one_tile(tile, gcur, g0, s0, g1, s1);
            one_tile(tile + t_stride, gcur, g1, s1, g0, s0);
        }
    } else {
        for (; tile < t_end; tile += 3 * t_stride) {         // branch-free body: slots past the end compute a clamped tile, store nothing
            one_tile(tile, gcur, g0, s0, g1, s1);
            one_tile(tile + t_stride, gcur, g1, s1, g2, s2);
            one_tile(tile + 2 * t_stride, gcur, g2, s2, g0, s0);
        }
    }
    if (STATS && ts.b >= 0) ts.flush(stat_part, sred);
}

// ------------------------------------------------------------------------------------------------
// The 48^3 level (32 -> 32 channels, 1728 tiles at B = 4): persistent k=3 kernel with the weights stationary in LDS.
//
// conv_k3_lds_kernel runs this shape as 1728 one-tile workgroups in 3.4 rounds of 2 per CU; every one of them pays an exposed
// halo round trip, three restages of the 55 KB weight tensor through registers and LDS with two barriers each, and an epilogue
// nothing overlaps.  Here ONE workgroup per CU
//   * copies the whole packed weight tensor (27 taps x 32 x 32 bf16 = 55 KB) into LDS once,
//   * walks ~7 tiles of its XCD's contiguous range with TWO halo images (38 KB each, the conflict-free rotated layout of
//     conv_k3_lds_kernel<32,..>): the next tile's halo is requested before the current tile's MFMAs and written into the other
//     image half-way through them -- one barrier per tile,
//   * runs the MFMAs transposed (A = weights, B = voxels): a lane then holds 4 consecutive output channels of one voxel and
//     stores them as 8 bytes straight from its accumulators -- no output staging through LDS,
//   * resolves the zero padding with one tile-uniform bit mask (valid z / y / x halo coordinates) against a per-piece constant:
//     boundary tiles (63 % of the tiles at 48^3) take the same path as inner ones.
// Per tile and wave: 27 k-steps of 4 + 2 ds_read_b128 for 8 MFMAs.  132 KB LDS, 1 workgroup per CU.
// ------------------------------------------------------------------------------------------------
constexpr int P32_VS = 32;                                   // LDS voxel stride in elements (64 B)
constexpr int P32_RP = CL_HX;                                // x-row pitch in voxels
constexpr int P32_XH = CL_HZ * CL_HY * P32_RP * P32_VS;      // one halo image, bf16 elements (38.4 KB)
constexpr int P32_BS = 27 * 2 * 64 * 8;                      // packed weights [tap][n-tile][lane][8] (55.3 KB)
#ifndef P32_LA
#define P32_LA 2                                             // LDS fragment reads run this many taps ahead of their MFMAs
#endif
#ifndef P32_BREG
#define P32_BREG 0                                           // 1: all 54 weight fragments in registers instead of LDS (measured: 32.8 vs 30.5 us)
#endif
#ifndef P32_SGB
#define P32_SGB 2                                            // VALU instructions requested between consecutive MFMAs (0: the compiler's order)
#endif
#ifndef P32_PIN
#define P32_PIN 1                                            // halo pieces pinned inside their taps (0: the compiler places them)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st8_untracked(void* p, unsigned lo, unsigned hi) {      // see st16_untracked
    const u32x2 q = {lo, hi};
    asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {                     // v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

#ifdef P32_STAMP       // diagnostic builds only (tools/p32_stamps.py): where workgroup 0 / wave 0 spends its cycles, summed over its tiles
__device__ unsigned long long p32_stamps[8];
#define P32_T(var) unsigned long long var; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0)
extern "C" int dycon_debug_p32_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(p32_stamps), sizeof(p32_stamps)); }
#else
#define P32_T(var)
#endif

struct P32Geo {             // wave-uniform description of one tile
    long long org;          // element offset of voxel (b, z0, y0, x0), 32 channels per voxel
    int z0, y0, x0;
    unsigned mask;          // bit hz: halo z valid | bit 6 + hy | bit 16 + hx
    int b;                  // sample
};

template <int NW, bool STATS>      // NW waves per workgroup: 4 (one per SIMD) or 8 (two per SIMD: a single wave is vector-issue bound); STATS: stat_part
__global__ __launch_bounds__(64 * NW, 1) void conv_k3_p32_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wf,
                                                             const float* __restrict__ bias, bf16* __restrict__ Y, int B, int D,
                                                             int H, int W, int tilesZ, int tilesY, int tilesX, int nTiles,
                                                             float* __restrict__ stat_part) {
    __shared__ __attribute__((aligned(16))) unsigned short Xh[2 * P32_XH];
#if !P32_BREG
    __shared__ __attribute__((aligned(16))) unsigned short Bs[P32_BS];
#endif
    int tile, t_end, t_stride;
    xcd_tile_range(nTiles, tile, t_end, t_stride);
    // stat_part: per-(sample, workgroup) {sum, sum of squares} of the STORED (bf16) outputs per channel, [B][gridDim.x][32][2] -- the
    // statistics pass of the normalisation that follows (norm_partial_kernel) then does not run: a persistent workgroup adds up its ~7
    // tiles in registers and writes one row per sample it touched (rows of the other samples: zero)
    if (STATS && threadIdx.x < 64)
        for (int n = 0; n < B; ++n) stat_part[((long long)n * gridDim.x + blockIdx.x) * 64 + threadIdx.x] = 0.f;
    if (tile >= t_end) return;                               // (uniform)
    constexpr int NTHR = 64 * NW, YH = NW / 4, MT = 4 / YH;   // y-halves of a z-slice over the waves, m-tiles (y-row pairs) per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int zs = wave & 3, yh = wave >> 2;                 // this wave: z-slice zs, y rows [8 / YH * yh, ...)
#ifdef P32_STAMP
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
    P32_T(k0);
#endif
#if P32_BREG
    // weights: all 54 B fragments (27 taps x 2 n-tiles) stay in this wave's registers for the lifetime of the workgroup -- with one
    // wave per SIMD the 512-entry register file has room, and the LDS array serves A fragments only (4 instead of 6 reads per tap)
    bf16x8 bw[27][2];
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            bw[t][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Wf + ((long long)(t * 2 + j) * 64 + lane) * 8));
#else
    // weights: every piece of this thread in flight at once (the packed order IS the fragment order: a linear copy)
    constexpr int NWP = P32_BS / 8;                          // 3456 pieces of 16 B
    constexpr int NWS = (NWP + NTHR - 1) / NTHR;             // 14 (7) per thread
    uint4 wst[NWS];
#pragma unroll
    for (int it = 0; it < NWS; ++it) wst[it] = *reinterpret_cast<const uint4*>(Wf + (long long)min((int)threadIdx.x + NTHR * it, NWP - 1) * 8);
#endif

    // A fragments (the MFMA's B operand): wave w owns z-slice w of the tile, m-tile m = y rows 2m, 2m+1; k-step t = tap t, lane group
    // kg reads the 8-channel chunk kg of voxel (x + dx), which sits at rotated position (kg + hx) & 3 of that voxel's 64 bytes.
    constexpr int MSTEP = 2 * P32_RP * P32_VS;
    int abase[3];
    {
        const int vb = ((zs * CL_HY + 2 * MT * yh + (r >> 3)) * P32_RP + (r & 7)) * P32_VS;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) abase[dx] = vb + dx * P32_VS + 8 * ((kg + (r & 7) + dx) & 3);
    }
    const int bbase = lane * 8;

    constexpr int NPC = CL_NH * 4;                           // 2400 halo pieces of 16 B
    constexpr int NST = (NPC + NTHR - 1) / NTHR;             // per thread (10 or 5; pieces past the end duplicate the last one)
    int rel[NST], lofs[NST];
    unsigned need[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int e = min((int)threadIdx.x + NTHR * it, NPC - 1);
        const int hv = e >> 2, pc = e & 3;
        const int hx = hv % CL_HX, hy = (hv / CL_HX) % CL_HY, hz = hv / (CL_HX * CL_HY);
        rel[it] = (((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * 32 + 8 * pc;
        lofs[it] = ((hv / CL_HX) * P32_RP + hx) * P32_VS + 8 * ((pc + hx) & 3);
        need[it] = (1u << hz) | (1u << (6 + hy)) | (1u << (16 + hx));
    }
    // output: lane (r, kg) holds channels 4 kg .. 4 kg + 3 (+16 for the second n-tile) of voxel r of each of its 4 m-tiles
    const int obase = ((zs * H + 2 * MT * yh + (r >> 3)) * W + (r & 7)) * 32 + 4 * kg;
    const int ostep = 2 * W * 32;
    float bv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[j][i] = bias ? bias[j * 16 + 4 * kg + i] : 0.f;

    auto geometry = [&](int t, P32Geo& g) {                  // scalar: tile index -> origin and validity mask (clamped past the end)
        t = min(t, t_end - 1);
        const int tx = t % tilesX; t /= tilesX;
        const int ty = t % tilesY; t /= tilesY;
        const int tz = t % tilesZ;
        const int b = t / tilesZ;
        g.z0 = tz * CL_TZ; g.y0 = ty * CL_TY; g.x0 = tx * CL_TX;
        g.org = ((((long long)b * D + g.z0) * H + g.y0) * W + g.x0) * 32;
        g.b = b;
        // valid halo coordinates h: 0 <= c0 + h - 1 < extent  <=>  h in [max(0, 1 - c0), min(n, extent - c0 + 1))
        auto bits = [](int c0, int extent, int n) {
            const int lo = c0 >= 1 ? 0 : 1, hi = min(n, extent - c0 + 1);
            return ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        };
        g.mask = bits(g.z0, D, CL_HZ) | (bits(g.y0, H, CL_HY) << 6) | (bits(g.x0, W, CL_HX) << 16);
    };
    // Halo loads are UNCONDITIONAL: a piece outside the volume reads the tile's origin voxel instead (a valid address) and is
    // zeroed when it is written to LDS, so every wave issues exactly NST loads per tile (counted vmcnt waits)
    auto load_piece = [&](const P32Geo& g, int it, uint4 (&stg)[NST]) {
        const bool in = (g.mask & need[it]) == need[it];
        stg[it] = *reinterpret_cast<const uint4*>(X + g.org + (in ? rel[it] : 0));
    };
    auto store_piece = [&](const P32Geo& g, int it, const uint4 (&stg)[NST], unsigned short* img) {
        const bool in = (g.mask & need[it]) == need[it];
        uint4 v = stg[it];
        if (!in) v = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(img + lofs[it]) = v;
    };

    float st1[2][4], st2[2][4];                              // this lane's channels 16 j + 4 kg + i, summed over its voxels of the current sample
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) st1[j][i] = st2[j][i] = 0.f;
    int stat_b = -1;
    __shared__ float sred[NW * 64];
    auto flush_stats = [&](int bsample) {                    // uniform call: all threads, between two tiles
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = st1[j][i], q2 = st2[j][i];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); q2 += __shfl_xor(q2, o, 64); }
                if (r == 0) { sred[(wave * 32 + 16 * j + 4 * kg + i) * 2] = a; sred[(wave * 32 + 16 * j + 4 * kg + i) * 2 + 1] = q2; }
                st1[j][i] = st2[j][i] = 0.f;
            }
        __syncthreads();
        if (threadIdx.x < 64) {
            float v = 0.f;
            for (int w = 0; w < NW; ++w) v += sred[w * 64 + threadIdx.x];
            stat_part[((long long)bsample * gridDim.x + blockIdx.x) * 64 + threadIdx.x] = v;
        }
        __syncthreads();
    };

    // One tile.  Halo pieces of the two tiles ahead travel in two register sets: set `ld` receives tile n+2 during taps 0..9 (one
    // piece per tap), set `st` -- requested a whole tile ago -- is written into the other image during taps 10..19.  Each piece
    // is pinned inside its tap (sched_barrier), so that its address arithmetic, its load issue or LDS write fills the
    // vector-issue slots the tap's 8 MFMAs leave free: with one wave per SIMD nothing else would.
    auto one_tile = [&](int cur, const P32Geo& gcur, const P32Geo& gst, const uint4 (&st)[NST], P32Geo& gld, uint4 (&ld)[NST], int buf) {
        P32_T(t0);
        const unsigned short* xh = Xh + buf * P32_XH;
        unsigned short* xo = Xh + (buf ^ 1) * P32_XH;
        geometry(cur + 2 * t_stride, gld);                   // (past the end: the last tile again -- loaded and written, never used)
        f32x4 acc[MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m) { acc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        constexpr int LA = P32_LA, RING = P32_LA + 1;        // fragments of tap t+LA are requested before the MFMAs of tap t
        bf16x8 afr[RING][MT];
#if !P32_BREG
        bf16x8 bfr[RING][2];
#endif
        auto fetch = [&](int n) {
            const int imm = ((n / 9) * CL_HY + (n / 3) % 3) * P32_RP * P32_VS;
#pragma unroll
            for (int m = 0; m < MT; ++m)
                afr[n % RING][m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xh + abase[n % 3] + imm + m * MSTEP));
#if !P32_BREG
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bfr[n % RING][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + bbase + (n * 2 + j) * 512));
#endif
        };
#pragma unroll
        for (int n = 0; n < LA; ++n) fetch(n);
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            if (t + LA < 27) fetch(t + LA);
            if (t < NST) load_piece(gld, t, ld);
            else if (t < 2 * NST) store_piece(gst, t - NST, st, xo);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j)   // transposed: D[cout 4 kg + i][voxel r]
#if P32_BREG
                    acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[t][j], afr[t % RING][m], acc[m][j], 0, 0, 0);
#else
                    acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[t % RING][j], afr[t % RING][m], acc[m][j], 0, 0, 0);
#endif
#if P32_SGB      // ask for MFMA / LDS read / 2 VALU in turn: an in-order wave hides other work only in the 8 issue cycles an MFMA leaves free
#pragma unroll
            for (int gq = 0; gq < MT * 2; ++gq) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, P32_SGB, 0);
            }
#endif
#if P32_PIN
            if (t < 2 * NST) __builtin_amdgcn_sched_barrier(0);
#endif
        }
        P32_T(t4);
        {
            bf16* yb = Y + gcur.org + obase;
            const bool zok = gcur.z0 + zs < D, xok = gcur.x0 + (r & 7) < W;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (zok && xok && gcur.y0 + 2 * MT * yh + 2 * m + (r >> 3) < H) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const unsigned lo = pack_bf16x2(acc[m][j][0] + bv[j][0], acc[m][j][1] + bv[j][1]);
                        const unsigned hi = pack_bf16x2(acc[m][j][2] + bv[j][2], acc[m][j][3] + bv[j][3]);
                        st8_untracked(yb + m * ostep + 16 * j, lo, hi);
                        if (STATS) {          // (uniform) statistics of the values as stored
                            const float v0 = __uint_as_float(lo << 16), v1 = __uint_as_float(lo & 0xffff0000u);
                            const float v2 = __uint_as_float(hi << 16), v3 = __uint_as_float(hi & 0xffff0000u);
                            st1[j][0] += v0; st2[j][0] += v0 * v0; st1[j][1] += v1; st2[j][1] += v1 * v1;
                            st1[j][2] += v2; st2[j][2] += v2 * v2; st1[j][3] += v3; st2[j][3] += v3 * v3;
                        }
                    }
                }
            }
        }
        P32_T(t5);
        __syncthreads();                                     // this tile's image consumed, the next one complete
        P32_T(t6);
#ifdef P32_STAMP
        seg[0] += t4 - t0; seg[3] += t5 - t4; seg[4] += t6 - t5;
#endif
    };

    uint4 sa[NST], sb[NST];
    P32Geo g0, g1, g2;
    geometry(tile, g0);
#pragma unroll
    for (int it = 0; it < NST; ++it) load_piece(g0, it, sa);
    geometry(tile + t_stride, g1);
#pragma unroll
    for (int it = 0; it < NST; ++it) load_piece(g1, it, sb);
#if !P32_BREG
#pragma unroll
    for (int it = 0; it < NWS; ++it)
        if ((int)threadIdx.x + NTHR * it < NWP) *reinterpret_cast<uint4*>(Bs + (threadIdx.x + NTHR * it) * 8) = wst[it];
#endif
#pragma unroll
    for (int it = 0; it < NST; ++it) store_piece(g0, it, sa, Xh);
    __syncthreads();
#ifdef P32_STAMP
    P32_T(k1);
#endif
    for (; tile < t_end; tile += 2 * t_stride) {             // tile j of this workgroup's sequence: image j % 2, store set (j+1) % 2
        if (STATS && g0.b != stat_b) { if (stat_b >= 0) flush_stats(stat_b); stat_b = g0.b; }
        one_tile(tile, g0, g1, sb, g2, sa, 0);               // computes g0, writes g1's halo (sb), requests g2 into sa
        if (tile + t_stride >= t_end) break;                 // (uniform)
        if (STATS && g1.b != stat_b) { flush_stats(stat_b); stat_b = g1.b; }
        one_tile(tile + t_stride, g1, g2, sa, g0, sb, 1);    // computes g1, writes g2's halo (sa), requests the next g0 into sb
        g1 = g0;                                             // rotate: the tile just requested is the one after the next
        g0 = g2;
        // after the swap: g0 = the tile to compute, whose halo is in image 0 -- its data travelled in sa; g1 = requested into sb
    }
    if (STATS && stat_b >= 0) flush_stats(stat_b);
#ifdef P32_STAMP
    P32_T(k2);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int i = 0; i < 6; ++i) p32_stamps[i] = seg[i];
        p32_stamps[6] = k1 - k0;
        p32_stamps[7] = k2 - k1;
    }
#endif
}

// The same kernel on v_mfma_f32_32x32x16_bf16: conv_k3_p32_kernel's tap loop is vector-ISSUE bound (stamps: ~215
// issue cycles per tap against 128 of MFMA at one wave per SIMD), and a 16x16x32 MFMA holds the SIMD's vector issue for 8 of its 16
// cycles, a 32x32x16 for 8 of its 32 (MI355X_MICROARCH.md, cycle constants): per tap a wave issues 2 (4) MFMAs instead of 4 (8) for
// the same 32 output channels x 32 (64) voxels, with the same number of ds_read_b128 (one weight and one activation fragment per
// k-step of 16 channels).  Output tile = 32 channels x 32 voxels (4 y rows x 8 x): one accumulator of 16 registers per 32 voxels.
// Differences to conv_k3_p32_kernel: fragment order of the weights in LDS (re-ordered in the prologue copy), rotation of the 8-channel
// chunks inside a voxel's 64 bytes (conflict-free for the 32-voxel read pattern), epilogue lane map.  Same results bit for bit is NOT
// expected (the k order inside a tap differs: two k-steps of 16 instead of one of 32): tested element-wise like every conv kernel.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NW, bool STATS>
__global__ __launch_bounds__(64 * NW, 1) void conv_k3_p32x_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wf,
                                                             const float* __restrict__ bias, bf16* __restrict__ Y, int B, int D,
                                                             int H, int W, int tilesZ, int tilesY, int tilesX, int nTiles,
                                                             float* __restrict__ stat_part) {
    __shared__ __attribute__((aligned(16))) unsigned short Xh[2 * P32_XH];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[P32_BS];
    int tile, t_end, t_stride;
    xcd_tile_range(nTiles, tile, t_end, t_stride);
    // stat_part: per-(sample, workgroup) {sum, sum of squares} of the STORED (bf16) outputs per channel, [B][gridDim.x][32][2] -- the
    // statistics pass of the normalisation that follows (norm_partial_kernel) then does not run: a persistent workgroup adds up its ~7
    // tiles in registers and writes one row per sample it touched (rows of the other samples: zero)
    if (STATS && threadIdx.x < 64)
        for (int n = 0; n < B; ++n) stat_part[((long long)n * gridDim.x + blockIdx.x) * 64 + threadIdx.x] = 0.f;
    if (tile >= t_end) return;                               // (uniform)
    constexpr int NTHR = 64 * NW, YH = NW / 4, NA = 2 / YH;   // y-halves of a z-slice over the waves; 32-voxel accumulators (4 y rows x 8) per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r5 = lane & 31, hl = lane >> 5;                // MFMA 32x32x16: column (voxel) r5, k half hl
    const int zs = wave & 3, yh = wave >> 2;                 // this wave: z-slice zs, y rows [8 / YH * yh, ...)
#ifdef P32_STAMP
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
    P32_T(k0);
#endif
    // weights: every piece of this thread in flight at once.  The packed order (dycon_pack_bfrag: [tap][16-column tile][lane of the
    // 16x16x32 fragment][8]) is re-ordered on the way into LDS into the A fragments of the 32x32x16 instruction, [tap][k-step of 16
    // channels][lane][8] with lane = (k half hl, output channel r5): a permutation of whole 16-byte pieces
    constexpr int NWP = P32_BS / 8;                          // 3456 pieces of 16 B
    constexpr int NWS = (NWP + NTHR - 1) / NTHR;             // 14 (7) per thread
    uint4 wst[NWS];
#pragma unroll
    for (int it = 0; it < NWS; ++it) {
        const int e = min((int)threadIdx.x + NTHR * it, NWP - 1);
        const int tp = e >> 7, ks = (e >> 6) & 1, hh = (e >> 5) & 1, rr = e & 31;
        const int src = tp * 128 + (rr >> 4) * 64 + ((2 * ks + hh) << 4) + (rr & 15);
        wst[it] = *reinterpret_cast<const uint4*>(Wf + (long long)src * 8);
    }

    // Activation fragments (the MFMA's B operand, 16 channels x 32 voxels): wave w owns z-slice zs, accumulator a = y rows 4a .. 4a+3
    // of its half; k-step ks of tap t reads the 8-channel chunk c = 2 ks + hl of voxel (y, x + dx), which sits at rotated position
    // (c + ((x + dx) >> 1)) & 3 of that voxel's 64 bytes: the 16 lanes ds_read_b128 services together then hit 64 distinct banks
    // (lane groups {0-3, 12-15, 20-27} ...: per 16-bank window the four voxels have x = w, w + 2, w + 4, w + 6 modulo 8).
    constexpr int ASTEP = 4 * P32_RP * P32_VS;
    int abase[3][2];
    {
        const int vb = ((zs * CL_HY + 4 * NA * yh + (r5 >> 3)) * P32_RP + (r5 & 7)) * P32_VS;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) abase[dx][ks] = vb + dx * P32_VS + 8 * ((2 * ks + hl + (((r5 & 7) + dx) >> 1)) & 3);
    }
    const int bbase = lane * 8;

    constexpr int NPC = CL_NH * 4;                           // 2400 halo pieces of 16 B
    constexpr int NST = (NPC + NTHR - 1) / NTHR;             // per thread (10 or 5; pieces past the end duplicate the last one)
    int rel[NST], lofs[NST];
    unsigned need[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int e = min((int)threadIdx.x + NTHR * it, NPC - 1);
        const int hv = e >> 2, pc = e & 3;
        const int hx = hv % CL_HX, hy = (hv / CL_HX) % CL_HY, hz = hv / (CL_HX * CL_HY);
        rel[it] = (((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * 32 + 8 * pc;
        lofs[it] = ((hv / CL_HX) * P32_RP + hx) * P32_VS + 8 * ((pc + (hx >> 1)) & 3);
        need[it] = (1u << hz) | (1u << (6 + hy)) | (1u << (16 + hx));
    }
    // output (C/D of 32x32x16): lane (r5, hl) holds, for its voxel r5 of each accumulator, channels 8 g + 4 hl + i  (register 4 g + i)
    const int obase = ((zs * H + 4 * NA * yh + (r5 >> 3)) * W + (r5 & 7)) * 32 + 4 * hl;
    const int ostep = 4 * W * 32;
    float bv[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[g][i] = bias ? bias[8 * g + 4 * hl + i] : 0.f;

    auto geometry = [&](int t, P32Geo& g) {                  // scalar: tile index -> origin and validity mask (clamped past the end)
        t = min(t, t_end - 1);
        const int tx = t % tilesX; t /= tilesX;
        const int ty = t % tilesY; t /= tilesY;
        const int tz = t % tilesZ;
        const int b = t / tilesZ;
        g.z0 = tz * CL_TZ; g.y0 = ty * CL_TY; g.x0 = tx * CL_TX;
        g.org = ((((long long)b * D + g.z0) * H + g.y0) * W + g.x0) * 32;
        g.b = b;
        // valid halo coordinates h: 0 <= c0 + h - 1 < extent  <=>  h in [max(0, 1 - c0), min(n, extent - c0 + 1))
        auto bits = [](int c0, int extent, int n) {
            const int lo = c0 >= 1 ? 0 : 1, hi = min(n, extent - c0 + 1);
            return ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        };
        g.mask = bits(g.z0, D, CL_HZ) | (bits(g.y0, H, CL_HY) << 6) | (bits(g.x0, W, CL_HX) << 16);
    };
    // Halo loads are UNCONDITIONAL: a piece outside the volume reads the tile's origin voxel instead (a valid address) and is
    // zeroed when it is written to LDS, so every wave issues exactly NST loads per tile (counted vmcnt waits)
    auto load_piece = [&](const P32Geo& g, int it, uint4 (&stg)[NST]) {
        const bool in = (g.mask & need[it]) == need[it];
        stg[it] = *reinterpret_cast<const uint4*>(X + g.org + (in ? rel[it] : 0));
    };
    auto store_piece = [&](const P32Geo& g, int it, const uint4 (&stg)[NST], unsigned short* img) {
        const bool in = (g.mask & need[it]) == need[it];
        uint4 v = stg[it];
        if (!in) v = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(img + lofs[it]) = v;
    };

    float st1[4][4], st2[4][4];                              // this lane's channels 8 g + 4 hl + i, summed over its voxels of the current sample
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) st1[g][i] = st2[g][i] = 0.f;
    int stat_b = -1;
    __shared__ float sred[NW * 64];
    auto flush_stats = [&](int bsample) {                    // uniform call: all threads, between two tiles
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = st1[g][i], q2 = st2[g][i];
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { a += __shfl_xor(a, o, 64); q2 += __shfl_xor(q2, o, 64); }
                if (r5 == 0) { sred[(wave * 32 + 8 * g + 4 * hl + i) * 2] = a; sred[(wave * 32 + 8 * g + 4 * hl + i) * 2 + 1] = q2; }
                st1[g][i] = st2[g][i] = 0.f;
            }
        __syncthreads();
        if (threadIdx.x < 64) {
            float v = 0.f;
            for (int w = 0; w < NW; ++w) v += sred[w * 64 + threadIdx.x];
            stat_part[((long long)bsample * gridDim.x + blockIdx.x) * 64 + threadIdx.x] = v;
        }
        __syncthreads();
    };

    // One tile.  Halo pieces of the two tiles ahead travel in two register sets: set `ld` receives tile n+2 during taps 0..9 (one
    // piece per tap), set `st` -- requested a whole tile ago -- is written into the other image during taps 10..19.  Each piece
    // is pinned inside its tap (sched_barrier), so that its address arithmetic, its load issue or LDS write fills the
    // vector-issue slots the tap's 8 MFMAs leave free: with one wave per SIMD nothing else would.
    auto one_tile = [&](int cur, const P32Geo& gcur, const P32Geo& gst, const uint4 (&st)[NST], P32Geo& gld, uint4 (&ld)[NST], int buf) {
        P32_T(t0);
        const unsigned short* xh = Xh + buf * P32_XH;
        unsigned short* xo = Xh + (buf ^ 1) * P32_XH;
        geometry(cur + 2 * t_stride, gld);                   // (past the end: the last tile again -- loaded and written, never used)
        f32x16 acc[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
        constexpr int LA = P32_LA, RING = P32_LA + 1;        // fragments of tap t+LA are requested before the MFMAs of tap t
        bf16x8 xfr[RING][2][NA], wfr[RING][2];
        auto fetch = [&](int n) {
            const int imm = ((n / 9) * CL_HY + (n / 3) % 3) * P32_RP * P32_VS;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int a = 0; a < NA; ++a)
                    xfr[n % RING][ks][a] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xh + abase[n % 3][ks] + imm + a * ASTEP));
                wfr[n % RING][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + bbase + (n * 2 + ks) * 512));
            }
        };
#pragma unroll
        for (int n = 0; n < LA; ++n) fetch(n);
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            if (t + LA < 27) fetch(t + LA);
            if (t < NST) load_piece(gld, t, ld);
            else if (t < 2 * NST) store_piece(gst, t - NST, st, xo);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < NA; ++a)     // transposed: D[cout][voxel r5] += W[cout][16 channels] . X[16 channels][voxel]
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[t % RING][ks], xfr[t % RING][ks][a], acc[a], 0, 0, 0);
#if P32_SGB      // ask for MFMA / LDS reads / VALU in turn: an in-order wave hides other work only in the issue cycles an MFMA leaves free
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {                 // per tap: 2 NA MFMAs, 2 NA + 2 fragment reads
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2 * P32_SGB, 0);
            }
            if (NA == 2) {
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2 * P32_SGB, 0);
                }
            }
#endif
#if P32_PIN
            if (t < 2 * NST) __builtin_amdgcn_sched_barrier(0);
#endif
        }
        P32_T(t4);
        {
            bf16* yb = Y + gcur.org + obase;
            const bool zok = gcur.z0 + zs < D, xok = gcur.x0 + (r5 & 7) < W;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                if (zok && xok && gcur.y0 + 4 * (NA * yh + a) + (r5 >> 3) < H) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const unsigned lo = pack_bf16x2(acc[a][4 * g] + bv[g][0], acc[a][4 * g + 1] + bv[g][1]);
                        const unsigned hi = pack_bf16x2(acc[a][4 * g + 2] + bv[g][2], acc[a][4 * g + 3] + bv[g][3]);
                        st8_untracked(yb + a * ostep + 8 * g, lo, hi);
                        if (STATS) {          // (uniform) statistics of the values as stored
                            const float v0 = __uint_as_float(lo << 16), v1 = __uint_as_float(lo & 0xffff0000u);
                            const float v2 = __uint_as_float(hi << 16), v3 = __uint_as_float(hi & 0xffff0000u);
                            st1[g][0] += v0; st2[g][0] += v0 * v0; st1[g][1] += v1; st2[g][1] += v1 * v1;
                            st1[g][2] += v2; st2[g][2] += v2 * v2; st1[g][3] += v3; st2[g][3] += v3 * v3;
                        }
                    }
                }
            }
        }
        P32_T(t5);
        __syncthreads();                                     // this tile's image consumed, the next one complete
        P32_T(t6);
#ifdef P32_STAMP
        seg[0] += t4 - t0; seg[3] += t5 - t4; seg[4] += t6 - t5;
#endif
    };

    uint4 sa[NST], sb[NST];
    P32Geo g0, g1, g2;
    geometry(tile, g0);
#pragma unroll
    for (int it = 0; it < NST; ++it) load_piece(g0, it, sa);
    geometry(tile + t_stride, g1);
#pragma unroll
    for (int it = 0; it < NST; ++it) load_piece(g1, it, sb);
#pragma unroll
    for (int it = 0; it < NWS; ++it)
        if ((int)threadIdx.x + NTHR * it < NWP) *reinterpret_cast<uint4*>(Bs + (threadIdx.x + NTHR * it) * 8) = wst[it];
#pragma unroll
    for (int it = 0; it < NST; ++it) store_piece(g0, it, sa, Xh);
    __syncthreads();
#ifdef P32_STAMP
    P32_T(k1);
#endif
    for (; tile < t_end; tile += 2 * t_stride) {             // tile j of this workgroup's sequence: image j % 2, store set (j+1) % 2
        if (STATS && g0.b != stat_b) { if (stat_b >= 0) flush_stats(stat_b); stat_b = g0.b; }
        one_tile(tile, g0, g1, sb, g2, sa, 0);               // computes g0, writes g1's halo (sb), requests g2 into sa
        if (tile + t_stride >= t_end) break;                 // (uniform)
        if (STATS && g1.b != stat_b) { flush_stats(stat_b); stat_b = g1.b; }
        one_tile(tile + t_stride, g1, g2, sa, g0, sb, 1);    // computes g1, writes g2's halo (sa), requests the next g0 into sb
        g1 = g0;                                             // rotate: the tile just requested is the one after the next
        g0 = g2;
        // after the swap: g0 = the tile to compute, whose halo is in image 0 -- its data travelled in sa; g1 = requested into sb
    }
    if (STATS && stat_b >= 0) flush_stats(stat_b);
#ifdef P32_STAMP
    P32_T(k2);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int i = 0; i < 6; ++i) p32_stamps[i] = seg[i];
        p32_stamps[6] = k1 - k0;
        p32_stamps[7] = k2 - k1;
    }
#endif
}

// First layer (ONE input channel -> 16 * NT): K = 27 taps, padded to a single 32-wide k-step whose A fragment is gathered
// from a 1.2 KB scalar halo image (8 ds_read_u16 per lane and m-tile).  One MFMA per 16 voxels and n-tile: the kernel is a
// pure HBM stream of its output (2 B in, 32 * NT B out per voxel).  wfrag = dycon_pack_bfrag(T = 27, Cin = 1).
template <int NT, bool STATS = false>
__global__ __launch_bounds__(256) void conv_k3_c1_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wf,
                                                         const float* __restrict__ bias, bf16* __restrict__ Y, int B, int D, int H,
                                                         int W, int tilesZ, int tilesY, int tilesX, int nTiles, int accumulate,
                                                         float* __restrict__ stat_part = nullptr) {
    constexpr int XP = 12;                                   // x-row pitch of the scalar image (elements)
    constexpr int CB = NT * 16, OS = CB + 8;
    __shared__ unsigned short Xs[CL_HZ * CL_HY * XP];
    __shared__ __attribute__((aligned(16))) unsigned short Ot[CL_NV * OS];
    __shared__ float sred[STATS ? 4 * NT * 32 : 1];
    if (STATS && (int)threadIdx.x < NT * 32)                 // rows of the samples this workgroup never touches: zero
        for (int n = 0; n < B; ++n) stat_part[((long long)n * gridDim.x + blockIdx.x) * (NT * 32) + threadIdx.x] = 0.f;
    TileStats<NT> ts;
    ts.clear();
    ts.b = -1;
    const unsigned short* Xg = reinterpret_cast<const unsigned short*>(X);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    bf16x8 bfr[NT];
    float bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Wf + ((long long)j * 64 + lane) * 8));
        bv[j] = bias ? bias[j * 16 + r] : 0.f;
    }
    int vbase[4], toffs[8];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int v = (wave * 4 + m) * 16 + r;
        vbase[m] = ((v >> 6) * CL_HY + ((v >> 3) & 7)) * XP + (v & 7);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {                            // this lane's taps 8 kg + e (taps >= 27: zero weights)
        int t = 8 * kg + e;
        if (t > 26) t = 26;
        toffs[e] = ((t / 9) * CL_HY + (t / 3) % 3) * XP + t % 3;
    }
    constexpr int NST = (CL_NH + 255) / 256;                 // 3 scalars per thread
    unsigned short stg[NST];
    auto tile_origin = [&](int tile, int& b, int& z0, int& y0, int& x0) {
        const int tx = tile % tilesX; tile /= tilesX;
        const int ty = tile % tilesY; tile /= tilesY;
        const int tz = tile % tilesZ;
        b = tile / tilesZ;
        z0 = tz * CL_TZ; y0 = ty * CL_TY; x0 = tx * CL_TX;
    };
    auto load_halo = [&](int tile) {
        int b, z0, y0, x0;
        tile_origin(tile, b, z0, y0, x0);
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int hv = threadIdx.x + 256 * it;
            const int hx = hv % CL_HX, hy = (hv / CL_HX) % CL_HY, hz = hv / (CL_HX * CL_HY);
            const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
            stg[it] = 0;
            if (hv < CL_NH && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                stg[it] = Xg[(((long long)b * D + z) * H + y) * W + x];
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int hv = threadIdx.x + 256 * it;
            if (hv < CL_NH) Xs[(hv / CL_HX) * XP + hv % CL_HX] = stg[it];
        }
    };

    int tile, t_end, t_stride;
    xcd_tile_range(nTiles, tile, t_end, t_stride);
    if (tile < t_end) { load_halo(tile); store_halo(); }
    __syncthreads();
    for (; tile < t_end; tile += t_stride) {
        const bool has_next = tile + t_stride < t_end;
        if (has_next) load_halo(tile + t_stride);
        f32x4 acc[4][NT];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w[e] = (unsigned)Xs[vbase[m] + toffs[2 * e]] | ((unsigned)Xs[vbase[m] + toffs[2 * e + 1]] << 16);
            const bf16x8 afr = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[j], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        __syncthreads();
        int b, z0, y0, x0;
        tile_origin(tile, b, z0, y0, x0);
        if (STATS && b != ts.b) { if (ts.b >= 0) ts.flush(stat_part, sred); ts.b = b; }      // (uniform)
        const bool full = z0 + CL_TZ <= D && y0 + CL_TY <= H && x0 + CL_TX <= W;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned short bits = f32_to_bf16_bits(acc[m][j][i] + bv[j]);
                    Ot[((wave * 4 + m) * 16 + 4 * kg + i) * OS + j * 16 + r] = bits;
                    if (STATS) {
                        const int vy = 2 * m + ((4 * kg + i) >> 3), vx = (4 * kg + i) & 7;
                        ts.add(j, bits, full || (z0 + wave < D && y0 + vy < H && x0 + vx < W));
                    }
                }
        if (has_next) store_halo();
        __syncthreads();
        constexpr int PPR = CB / 8;
#pragma unroll
        for (int it = 0; it < PPR; ++it) {
            const int e = threadIdx.x + 256 * it;
            const int v = e / PPR, pc = e % PPR;
            const int z = z0 + (v >> 6), y = y0 + ((v >> 3) & 7), x = x0 + (v & 7);
            if (z >= D || y >= H || x >= W) continue;
            bf16* yp = Y + ((((long long)b * D + z) * H + y) * W + x) * CB + 8 * pc;
            Vec16<bf16> o;
            o.v = *reinterpret_cast<const uint4*>(Ot + v * OS + 8 * pc);
            if (accumulate) {
                const Vec16<bf16> old = ld16(yp);
#pragma unroll
                for (int k = 0; k < 8; ++k) o.set(k, o.get(k) + old.get(k));
            }
            st16(yp, o);
        }
    }
    if (STATS && ts.b >= 0) ts.flush(stat_part, sred);
}

// y[m, n] (+)= bias[n] + sum_z slab[z][m][n]   (ordered: deterministic)
// Four consecutive outputs per thread (N % 16 == 0, so a quad never straddles a row): one 16-byte load per slab, the slabs of a
// thread all in flight.  This launch follows every split-K convolution on the step's dependent chain.
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slab, int splits, long long MN, int N,
                                                            const float* __restrict__ bias, T* __restrict__ Y, int accumulate) {
    const long long nq = MN >> 2;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long long)gridDim.x * 256) {
        const long long i = q << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) v = *reinterpret_cast<const float4*>(bias + i % N);
        for (int z = 0; z < splits; ++z) {        // (ordered: the same summation order as before, per element)
            const float4 p = *reinterpret_cast<const float4*>(slab + (long long)z * MN + i);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if constexpr (sizeof(T) == 2) {
            if (accumulate) {
                const uint2 o = *reinterpret_cast<const uint2*>(Y + i);
                v.x += bf16_bits_to_f32((unsigned short)(o.x & 0xffffu)); v.y += bf16_bits_to_f32((unsigned short)(o.x >> 16));
                v.z += bf16_bits_to_f32((unsigned short)(o.y & 0xffffu)); v.w += bf16_bits_to_f32((unsigned short)(o.y >> 16));
            }
            uint2 o;
            o.x = (unsigned)f32_to_bf16_bits(v.x) | ((unsigned)f32_to_bf16_bits(v.y) << 16);
            o.y = (unsigned)f32_to_bf16_bits(v.z) | ((unsigned)f32_to_bf16_bits(v.w) << 16);
            *reinterpret_cast<uint2*>(Y + i) = o;
        } else {
            if (accumulate) {
                const float4 o = *reinterpret_cast<const float4*>(Y + i);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *reinterpret_cast<float4*>(Y + i) = v;
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

// ------------------------------------------------------------------------------------------------
// k=3 weight gradient on the bf16 matrix cores.
//
//   dW[t][ci][co] = sum_voxels X[voxel + off(t)][ci] * G[voxel][co]        (K = voxels)
//
// Both MFMA operands need the voxel index contiguous per lane, the opposite of the NDHWC memory order, so
// tiles are staged through LDS in their natural [voxel][channel] order (16-byte global loads) and read
// back TRANSPOSED by ds_read_b64_tr_b16: each 16-lane group fetches 4 voxels x 16 channels and every
// lane receives its channel's 4 voxels.  One workgroup = one 16-channel ci tile x up to 64 co, walking
// 4x4x8-voxel tiles (halo 6x6x10); the four waves split the 27 taps, so each wave keeps 7 x NT
// accumulator tiles in registers across ALL its voxel tiles and nothing is reduced across waves.
// The bias gradient (column sums of G) rides along in wave 0 of the ci-tile-0 workgroups.
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr int WG_TZ = 4, WG_TY = 4, WG_TX = 8;                       // voxel tile
constexpr int WG_HZ = WG_TZ + 2, WG_HY = WG_TY + 2, WG_HX = WG_TX + 2;   // halo tile
constexpr int WG_NV = WG_TZ * WG_TY * WG_TX;                          // 128 voxels = 4 MFMA k-steps
constexpr int WG_NH = WG_HZ * WG_HY * WG_HX;                          // 360 halo voxels
#ifndef WG_PAD
#define WG_PAD 1     // 0: natural LDS pitches (A/B timing builds)
#endif
constexpr int WG_XP = WG_PAD ? 12 : WG_HX;                                             // x pitch of the halo's LDS image (voxels): see wgrad_k3_bf16_kernel

__device__ __forceinline__ bf16x8 tr_frag(const unsigned short* lds_lo, const unsigned short* lds_hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)lds_lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)lds_hi);
    const s16x8 t = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, t);
}

// NW = waves per workgroup: 4, or 8 where the grid gives a CU at most one workgroup (the 24^3 / 12^3 / 6^3 levels: <= 256 workgroups,
// i.e. ONE wave per SIMD with nothing to hide its global-load, LDS and barrier latencies behind -- PMC: 42 % of a wave's life in
// s_waitcnt / s_barrier, matrix pipe busy 21 %).  Eight waves split the 27 taps 4 / 3 instead of 7 / 6: half the accumulators per
// wave, half the staging work per thread, two waves per SIMD.
template <int NT, bool CIN1, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void wgrad_k3_bf16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ GY,
                                                            float* __restrict__ part, float* __restrict__ bias_part, int B, int D,
                                                            int H, int W, int Cin, int Cout, int nCoBlk, int nTiles, int tilesZ,
                                                            int tilesY, int tilesX) {
    constexpr int CB = 16 * NT;
    // LDS images laid out for the transposed reads (ds_read_b64_tr_b16: banks = dword address mod 64, lanes 0-31 and 32-63 served
    // separately; a lane group = 2 x-rows (kg) x 4 voxels (q) x 4 channel quads (p) of 8 bytes, so the 8 (kg, q) blocks must start
    // at 8 distinct multiples of 8 dwords mod 64).  Natural pitches put them on top of each other -- PMC had 44-57 % of the LDS
    // cycles as bank conflicts (profiles/r03_pmc_small_levels.txt):
    //   X halo: 8 dwords per voxel (q -> 0, 8, 16, 24); x pitch padded from 10 to 12 voxels = 96 dwords (kg -> +32)
    //   G tile: voxel stride GVS (NT = 4: 40 dwords, q -> 0, 40, 16, 56; else the natural 8 NT), row stride GRS = 8 GVS + pad = 32
    //           (NT = 2: 8) mod 64 dwords
    constexpr int GVS = WG_PAD && NT == 4 ? 80 : CB;             // elements
    constexpr int GRS = 8 * GVS + (!WG_PAD ? 0 : NT == 2 ? 16 : 64);
    __shared__ __attribute__((aligned(16))) unsigned short Xh[WG_HZ * WG_HY * WG_XP * 16];
    __shared__ __attribute__((aligned(16))) unsigned short Gt[(WG_NV / 8) * GRS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kg = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int ci0 = (blockIdx.x / nCoBlk) * 16;
    const int co0 = (blockIdx.x % nCoBlk) * CB;
    constexpr int NTHR = 64 * NW, TPW = (27 + NW - 1) / NW;      // taps per wave: wave, wave + NW, ...
    const int ntaps = (27 - wave + NW - 1) / NW;
    f32x4 acc[TPW][NT];
#pragma unroll
    for (int a = 0; a < TPW; ++a)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bsum[j] = 0.f;
    const bool do_bias = bias_part != nullptr && ci0 == 0 && wave == 0;

    // Register-staged pipeline over this workgroup's tiles: the global loads of tile i+1 are issued before the MFMAs of tile i
    // and written to LDS after the barrier that retires tile i (the X halo and G tile of the next tile are in flight while
    // the matrix cores work).  CIN1 (first layer, one input channel): channel 0 carries x, channels 1..15 are zero.
    constexpr int NSX = CIN1 ? (WG_NH + NTHR - 1) / NTHR : (WG_NH * 2 + NTHR - 1) / NTHR;
    constexpr int NSG = (WG_NV * (CB / 8) + NTHR - 1) / NTHR;
    uint4 sx[NSX], sg[NSG];
    auto load_tile = [&](int tile) {
        int rr = tile;
        const int tx = rr % tilesX; rr /= tilesX;
        const int ty = rr % tilesY; rr /= tilesY;
        const int tz = rr % tilesZ;
        const int b = rr / tilesZ;
        const int z0 = tz * WG_TZ, y0 = ty * WG_TY, x0 = tx * WG_TX;
#pragma unroll
        for (int it = 0; it < NSX; ++it) {
            const int e = threadIdx.x + NTHR * it;
            const int hv = CIN1 ? e : e >> 1, half = CIN1 ? 0 : e & 1;
            const int hx = hv % WG_HX, hy = (hv / WG_HX) % WG_HY, hz = hv / (WG_HX * WG_HY);
            const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
            sx[it] = make_uint4(0, 0, 0, 0);
            if (hv < WG_NH && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
                if (CIN1) sx[it].x = reinterpret_cast<const unsigned short*>(X)[(((long long)b * D + z) * H + y) * W + x];
                else sx[it] = *reinterpret_cast<const uint4*>(X + ((((long long)b * D + z) * H + y) * W + x) * Cin + ci0 + 8 * half);
            }
        }
#pragma unroll
        for (int it = 0; it < NSG; ++it) {
            const int e = threadIdx.x + NTHR * it;
            const int v8 = e % (CB / 8), vv = e / (CB / 8);
            const int vx = vv % WG_TX, vy = (vv / WG_TX) % WG_TY, vz = vv / (WG_TX * WG_TY);
            const int z = z0 + vz, y = y0 + vy, x = x0 + vx;
            sg[it] = make_uint4(0, 0, 0, 0);
            if (e < WG_NV * (CB / 8) && z < D && y < H && x < W && co0 + 8 * v8 < Cout)
                sg[it] = *reinterpret_cast<const uint4*>(GY + ((((long long)b * D + z) * H + y) * W + x) * Cout + co0 + 8 * v8);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < NSX; ++it) {
            const int e = threadIdx.x + NTHR * it;
            const int hv = CIN1 ? e : e >> 1;
            const int lv = (hv / WG_HX) * WG_XP + hv % WG_HX;     // halo voxel -> padded LDS voxel
            if (CIN1) {
                if (e < WG_NH) {
                    *reinterpret_cast<uint4*>(Xh + lv * 16) = sx[it];
                    *reinterpret_cast<uint4*>(Xh + lv * 16 + 8) = make_uint4(0, 0, 0, 0);
                }
            } else if (e < WG_NH * 2) {
                *reinterpret_cast<uint4*>(Xh + lv * 16 + 8 * (e & 1)) = sx[it];
            }
        }
#pragma unroll
        for (int it = 0; it < NSG; ++it) {
            const int e = threadIdx.x + NTHR * it;
            const int vv = e / (CB / 8);
            if (e < WG_NV * (CB / 8)) *reinterpret_cast<uint4*>(Gt + (vv >> 3) * GRS + (vv & 7) * GVS + 8 * (e % (CB / 8))) = sg[it];
        }
    };

    // every split walks ONE contiguous range of tiles (not a stride of gridDim.y): consecutive tiles share halo planes, which the
    // workgroup then finds in its XCD's L2
    const int t_per = (nTiles + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t_beg = blockIdx.y * t_per, t_end = min(nTiles, t_beg + t_per);
    if (t_beg < t_end) load_tile(t_beg);
    for (int tile = t_beg; tile < t_end; ++tile) {
        __syncthreads();   // previous tile fully consumed
        store_tile();
        __syncthreads();
        if (tile + 1 < t_end) load_tile(tile + 1);
        // ---- 4 k-steps of 32 voxels: lane group kg owns x-row (z, y) = ((4s+kg)>>2, (4s+kg)&3), voxels x = 0..7
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = 4 * s + kg, z = row >> 2, y = row & 3;
            bf16x8 bfr[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const unsigned short* g0 = Gt + row * GRS + q * GVS + 16 * j + 4 * p;
                bfr[j] = tr_frag(g0, g0 + 4 * GVS);
                if (do_bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[j] += (float)bfr[j][e];
                }
            }
#pragma unroll
            for (int a = 0; a < TPW; ++a) {
                if (a < ntaps) {
                    const int t = wave + NW * a;
                    const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
                    const unsigned short* a0 = Xh + (((z + dz) * WG_HY + (y + dy)) * WG_XP + q + dx) * 16 + 4 * p;
                    const bf16x8 afr = tr_frag(a0, a0 + 4 * 16);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[j], acc[a][j], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial[split][t][ci][co]; D tile: row (ci) = 4*kg + reg, col (co) = lane & 15
    const int col = lane & 15;
    float* dst = part + (long long)blockIdx.y * 27 * Cin * Cout;
#pragma unroll
    for (int a = 0; a < TPW; ++a) {
        if (a >= ntaps) continue;
        const int t = wave + NW * a;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = co0 + 16 * j + col;
            if (co >= Cout) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (ci0 + 4 * kg + i < Cin) dst[((long long)t * Cin + ci0 + 4 * kg + i) * Cout + co] = acc[a][j][i];
        }
    }
    if (do_bias) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float v = bsum[j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int co = co0 + 16 * j + col;
            if (kg == 0 && co < Cout) bias_part[(long long)blockIdx.y * Cout + co] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the FIRST layer (one input channel -> 16): dW[co][tap] = sum_v gy[v][co] * x[v + tap].
// It is the last launch of the step's backward (its input is the data gradient of block_one's normalisation) and nothing can
// overlap it, so its duration is step time.  wgrad_k3_bf16_kernel<1, true> treats the single channel as a 16-channel block
// (27 MFMAs per 32 voxels at 1/16 utilisation, two barriers per 128-voxel tile, one tile of prefetch: 91 us for 120 MB).
// Here the product is laid out the other way round: D[tap (32, 27 used)][co (16)] += Xp^T[tap][voxel] . G[voxel][co], TWO MFMAs per
// 32 voxels; the A fragment of a tap is 8 x-neighbours of the halo row, read with one ds_read_b128 from one of three copies of
// the (1.2 KB) halo pre-shifted by dx; persistent workgroups walk 4x8x8 tiles with the gy tiles of the next DEPTH tiles in flight
// in registers.  A pure stream of gy.
// ------------------------------------------------------------------------------------------------
#ifndef WC1_DEPTH
#define WC1_DEPTH 3
#endif
// NB ("norm backward on load"): GY is the gradient w.r.t. the OUTPUT of the normalisation that follows the first convolution, Z that
// normalisation's input; the data gradient gz of the normalisation -- whose only consumer is this weight gradient, the first
// convolution has no data gradient -- is formed per element while the tile is written to LDS (nb[] = per (sample, channel)
// {alpha, beta, gamma, p1, p0}: gz = alpha * [p1 z + p0 > 0] gy + beta z + gamma, rounded to bf16 as the stored gz would have been).
// The backward-apply pass of that normalisation (2 reads + 1 write of the step's largest tensor) and this kernel's read of gz go away.
struct C1NormBwd { const bf16* Z; const float* stats; const float* gamma; const float* beta; const float* ab; const float* chan_scale;
                   int Nb, G, relu; float inv_cnt; };
template <bool NB>
__global__ __launch_bounds__(256, 2) void wgrad_k3_c1_kernel(const bf16* __restrict__ X, const bf16* __restrict__ GY,
                                                             float* __restrict__ part, float* __restrict__ bias_part, int B, int D,
                                                             int H, int W, int tilesZ, int tilesY, int tilesX, int nTiles,
                                                             C1NormBwd nbp) {
    constexpr int NROW = CL_HZ * CL_HY;                        // 60 halo x-rows
    __shared__ __attribute__((aligned(16))) unsigned short Xs[3][NROW][8];        // halo rows shifted by dx: Xs[dx][row][e] = x[row][dx + e]
    __shared__ __attribute__((aligned(16))) unsigned short Gt[CL_NV * 16];        // gy tile, natural order [voxel][16]
    __shared__ float red[4][32 * 16 + 16];
    __shared__ __attribute__((aligned(16))) float nb[NB ? 16 : 1][5][16];          // [sample][constant][channel]
    if (NB) {
        for (int i = threadIdx.x; i < B * 16; i += 256) {
            const int bb = i >> 4, c = i & 15, n = nbp.Nb == 1 ? 0 : bb, g = c / (16 / nbp.G);
            const float mu = nbp.stats[((long long)n * nbp.G + g) * 2], rs = nbp.stats[((long long)n * nbp.G + g) * 2 + 1];
            const float gm = nbp.gamma ? nbp.gamma[c] : 1.f, bt = nbp.beta ? nbp.beta[c] : 0.f;
            const float pa = nbp.ab[((long long)n * nbp.G + g) * 2] * nbp.inv_cnt, pb = nbp.ab[((long long)n * nbp.G + g) * 2 + 1] * nbp.inv_cnt;
            const float dr = nbp.chan_scale ? nbp.chan_scale[(long long)n * 16 + c] : 1.f;
            // gz = rs * (gm * g' - (pa + xh * pb)),  xh = (z - mu) * rs,  g' = [gm * xh + bt > 0] * dr * gy
            nb[bb][0][c] = rs * gm * dr;
            nb[bb][1][c] = -rs * rs * pb;
            nb[bb][2][c] = rs * (rs * pb * mu - pa);
            nb[bb][3][c] = nbp.relu ? gm * rs : 0.f;
            nb[bb][4][c] = nbp.relu ? bt - gm * rs * mu : 1.f;
        }
        __syncthreads();
    }
    int tile, t_end, t_stride;
    xcd_tile_range(nTiles, tile, t_end, t_stride);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4, q = r >> 2, p = lane & 3;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float bsum = 0.f;
    // this lane's taps: r (first M tile) and 16 + r (second; >= 27: zero rows)
    int aoff[2];
    bool aok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int t = 16 * mt + r;
        aok[mt] = t < 27;
        const int tt = aok[mt] ? t : 0;
        const int dz = tt / 9, dy = (tt / 3) % 3, dx = tt % 3;
        aoff[mt] = (dx * NROW + dz * CL_HY + dy) * 8;
    }
    // halo elements of this thread (600 scalars over 256 threads) and its two 16-byte pieces of the gy tile
    constexpr int NSX = (CL_NH + 255) / 256;                    // 3
    int xrel[NSX], xrow[NSX], xhx[NSX];
    unsigned xneed[NSX];
#pragma unroll
    for (int it = 0; it < NSX; ++it) {
        const int e = min((int)threadIdx.x + 256 * it, CL_NH - 1);
        const int hx = e % CL_HX, hy = (e / CL_HX) % CL_HY, hz = e / (CL_HX * CL_HY);
        xrel[it] = ((hz - 1) * H + (hy - 1)) * W + (hx - 1);
        xrow[it] = hz * CL_HY + hy;
        xhx[it] = hx;
        xneed[it] = (1u << hz) | (1u << (6 + hy)) | (1u << (16 + hx));
    }
    int grel[2], gofs[2];
    unsigned gneed[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = threadIdx.x + 256 * it;
        const int v = e >> 1, pc = e & 1;
        const int vz = v >> 6, vy = (v >> 3) & 7, vx = v & 7;
        grel[it] = ((vz * H + vy) * W + vx) * 16 + 8 * pc;
        gofs[it] = v * 16 + 8 * pc;
        gneed[it] = (1u << (vz + 1)) | (1u << (6 + vy + 1)) | (1u << (16 + vx + 1));    // interior voxel = halo coordinate + 1
    }
    struct Geo { long long org; unsigned mask; int b; };
    auto geometry = [&](int t, Geo& g) {
        t = min(t, t_end - 1);
        const int tx = t % tilesX; t /= tilesX;
        const int ty = t % tilesY; t /= tilesY;
        const int tz = t % tilesZ;
        const int b = t / tilesZ;
        const int z0 = tz * CL_TZ, y0 = ty * CL_TY, x0 = tx * CL_TX;
        g.org = (((long long)b * D + z0) * H + y0) * W + x0;
        g.b = b;
        auto bits = [](int c0, int extent, int n) {
            const int lo = c0 >= 1 ? 0 : 1, hi = min(n, extent - c0 + 1);
            return ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        };
        g.mask = bits(z0, D, CL_HZ) | (bits(y0, H, CL_HY) << 6) | (bits(x0, W, CL_HX) << 16);
    };
    struct Stage { unsigned short x[NSX]; uint4 g[2]; uint4 z[NB ? 2 : 1]; };
    const unsigned short* Xu = reinterpret_cast<const unsigned short*>(X);
    auto load_tile = [&](const Geo& g, Stage& st) {            // unconditional loads (an invalid piece reads the tile's origin voxel)
#pragma unroll
        for (int it = 0; it < NSX; ++it) {
            const bool in = (g.mask & xneed[it]) == xneed[it];
            st.x[it] = Xu[g.org + (in ? xrel[it] : 0)];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const bool in = (g.mask & gneed[it]) == gneed[it];
            st.g[it] = *reinterpret_cast<const uint4*>(GY + g.org * 16 + (in ? grel[it] : 0));
            if (NB) st.z[it] = *reinterpret_cast<const uint4*>(nbp.Z + g.org * 16 + (in ? grel[it] : 0));
        }
    };
    auto store_tile = [&](const Geo& g, const Stage& st) {
#pragma unroll
        for (int it = 0; it < NSX; ++it) {
            if ((int)threadIdx.x + 256 * it < CL_NH) {
                const bool in = (g.mask & xneed[it]) == xneed[it];
                const unsigned short v = in ? st.x[it] : (unsigned short)0;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int e = xhx[it] - dx;
                    if (e >= 0 && e < 8) Xs[dx][xrow[it]][e] = v;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const bool in = (g.mask & gneed[it]) == gneed[it];
            uint4 v = st.g[it];
            if (NB) {                                           // gy, z -> gz for this piece's 8 channels of sample g.b
                const int c0 = 8 * ((threadIdx.x + 256 * it) & 1);
                Vec16<bf16> gv, zv, o;
                gv.v = st.g[it];
                zv.v = st.z[it];
                const float* tb = &nb[g.b][0][c0];
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    const float z = zv.get(k2);
                    float gg = gv.get(k2);
                    if (!(tb[3 * 16 + k2] * z + tb[4 * 16 + k2] > 0.f)) gg = 0.f;
                    o.set(k2, tb[0 * 16 + k2] * gg + (tb[1 * 16 + k2] * z + tb[2 * 16 + k2]));
                }
                v = o.v;
            }
            if (!in) v = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(Gt + gofs[it]) = v;
        }
    };
    auto compute = [&]() {                                      // wave w: z-slice w of the tile = 2 k-steps of 4 x-rows
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int row = wave * 8 + 4 * s + kg;              // x-row (z, y) of the tile, 8 voxels
            const unsigned short* g0 = Gt + (row * 8 + q) * 16 + 4 * p;
            const bf16x8 bfr = tr_frag(g0, g0 + 4 * 16);        // B[k = voxel x][n = channel r]
            if (bias_part != nullptr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum += (float)bfr[e];
            }
            const int hrow = (wave * CL_HY + 4 * s + kg) * 8;   // halo row (z + dz, y + dy) is hrow + (dz * HY + dy) * 8
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                uint4 a = *reinterpret_cast<const uint4*>(&Xs[0][0][0] + aoff[mt] + hrow);
                if (!aok[mt]) a = make_uint4(0, 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), bfr, acc[mt], 0, 0, 0);
            }
        }
    };

    if (tile < t_end) {
        Stage st[WC1_DEPTH];
        Geo gg[WC1_DEPTH];
#pragma unroll
        for (int d = 0; d < WC1_DEPTH; ++d) { geometry(tile + d * t_stride, gg[d]); load_tile(gg[d], st[d]); }
        // tile j of this workgroup travels in set j % DEPTH; the loop is unrolled DEPTH times so that the sets are compile-time
        for (; tile < t_end; tile += WC1_DEPTH * t_stride) {
#pragma unroll
            for (int d = 0; d < WC1_DEPTH; ++d) {
                if (tile + d * t_stride < t_end) {              // (uniform)
                    __syncthreads();                            // previous tile consumed
                    store_tile(gg[d], st[d]);
                    geometry(tile + (d + WC1_DEPTH) * t_stride, gg[d]);
                    load_tile(gg[d], st[d]);                    // DEPTH tiles ahead (clamped past the end: loaded, never stored)
                    __syncthreads();
                    compute();
                }
            }
        }
    }
    // ---- this workgroup's partial: D tile row (tap) = 4 kg + i, column (co) = r; summed over the four waves through LDS
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave][(16 * mt + 4 * kg + i) * 16 + r] = acc[mt][i];
    {
        float v = bsum;
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (kg == 0) red[wave][32 * 16 + r] = v;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 27 * 16; o += 256)            // part[split][t][ci = 0][co]
        part[(long long)blockIdx.x * 27 * 16 + o] = red[0][o] + red[1][o] + red[2][o] + red[3][o];
    if (bias_part != nullptr && threadIdx.x < 16) {
        const int o = 32 * 16 + threadIdx.x;
        bias_part[(long long)blockIdx.x * 16 + threadIdx.x] = red[0][o] + red[1][o] + red[2][o] + red[3][o];
    }
}

// ------------------------------------------------------------------------------------------------
// The whole backward of block_one (first convolution 1 -> 16, normalisation, ReLU; VNet.py:176) as ONE pass over (x, z, gy).
// The normalisation's data gradient is affine in its inputs once the forward statistics are known,
//     gz = alpha * g' + beta * z + gamma,      g' = [p1 z + p0 > 0] * dr * gy   (forward constants only),
// with alpha, beta, gamma per (sample, channel) functions of the group sums {sum g', sum g' z} that the SAME pass is taking.  So the
// first layer's weight gradient  dW[c][t] = sum_v gz[v][c] x[v + t]  is  alpha * GX + beta * ZX + gamma * SX  with three tap
// correlations  GX = sum g' x,  ZX = sum z x,  SX = sum x  that do not depend on those sums: one streaming pass produces, per
// (workgroup, sample), GX, ZX, SX and the per-channel sums {sum g', sum g' z, sum z}; a small reduce and a one-workgroup finalize
// turn them into dW, db, dgamma, dbeta.  Replaces the statistics pass + finalize of the normalisation's backward, the weight-gradient
// pass and its reduce: (x, z, gy) are read once instead of twice, and nothing of it runs on the weight-gradient stream.  All MFMA
// operands are exact in bf16 (g' = gy or 0 -- times the power-of-two dropout factor --, z, x), so dW carries no bf16 rounding of gz.
// Structure of wgrad_k3_c1_kernel: persistent workgroups, transposed product D[tap][column], tiles 3 deep in registers.
// ------------------------------------------------------------------------------------------------
constexpr int FB_GX = 0, FB_ZX = 27 * 16, FB_SX = 2 * 27 * 16, FB_SG = FB_SX + 32, FB_SGZ = FB_SG + 16, FB_SZ = FB_SGZ + 16;
constexpr int FB_ROW = FB_SZ + 16;                            // 944 floats per (sample, workgroup)
struct FbFwd { const float* stats; const float* gamma; const float* beta; const float* chan_scale; int Nb, G, relu; };

__global__ __launch_bounds__(256, 2) void first_block_bwd_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Z,
                                                                 const bf16* __restrict__ GY, float* __restrict__ part, int B, int D,
                                                                 int H, int W, int tilesZ, int tilesY, int tilesX, int nTiles, FbFwd f,
                                                                 double* __restrict__ tot) {
    constexpr int NROW = CL_HZ * CL_HY;
    if (blockIdx.x == 0)                                        // the reduce kernel that follows adds into tot
        for (int o = threadIdx.x; o < B * FB_ROW; o += 256) tot[o] = 0.0;
    __shared__ __attribute__((aligned(16))) unsigned short Xs[3][NROW][8];
    __shared__ __attribute__((aligned(16))) unsigned short Gt1[CL_NV * 16];       // g'
    __shared__ __attribute__((aligned(16))) unsigned short Gt2[CL_NV * 16];       // z
    __shared__ float red[4][FB_ROW];
    __shared__ __attribute__((aligned(16))) float tab[16][3][16];                 // [sample][p1 | p0 | dr][channel]
    for (int n = 0; n < B; ++n)
        for (int o = threadIdx.x; o < FB_ROW; o += 256) part[((long long)n * gridDim.x + blockIdx.x) * FB_ROW + o] = 0.f;
    for (int i = threadIdx.x; i < B * 16; i += 256) {
        const int bb = i >> 4, c = i & 15, n = f.Nb == 1 ? 0 : bb, g = c / (16 / f.G);
        const float mu = f.stats[((long long)n * f.G + g) * 2], rs = f.stats[((long long)n * f.G + g) * 2 + 1];
        const float gm = f.gamma ? f.gamma[c] : 1.f, bt = f.beta ? f.beta[c] : 0.f;
        tab[bb][0][c] = f.relu ? gm * rs : 0.f;
        tab[bb][1][c] = f.relu ? bt - gm * rs * mu : 1.f;
        tab[bb][2][c] = f.chan_scale ? f.chan_scale[(long long)n * 16 + c] : 1.f;
    }
    __syncthreads();
    int tile, t_end, t_stride;
    xcd_tile_range(nTiles, tile, t_end, t_stride);
    if (tile >= t_end) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4, q = r >> 2, p = lane & 3;
    f32x4 acc1[2], acc2[2], acc3[2];
    float sg = 0.f, sgz = 0.f, sz = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc1[mt] = acc2[mt] = acc3[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int aoff[2];
    bool aok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int t = 16 * mt + r;
        aok[mt] = t < 27;
        const int tt = aok[mt] ? t : 0;
        const int dz = tt / 9, dy = (tt / 3) % 3, dx = tt % 3;
        aoff[mt] = (dx * NROW + dz * CL_HY + dy) * 8;
    }
    constexpr int NSX = (CL_NH + 255) / 256;
    int xrel[NSX], xrow[NSX], xhx[NSX];
    unsigned xneed[NSX];
#pragma unroll
    for (int it = 0; it < NSX; ++it) {
        const int e = min((int)threadIdx.x + 256 * it, CL_NH - 1);
        const int hx = e % CL_HX, hy = (e / CL_HX) % CL_HY, hz = e / (CL_HX * CL_HY);
        xrel[it] = ((hz - 1) * H + (hy - 1)) * W + (hx - 1);
        xrow[it] = hz * CL_HY + hy;
        xhx[it] = hx;
        xneed[it] = (1u << hz) | (1u << (6 + hy)) | (1u << (16 + hx));
    }
    int grel[2], gofs[2];
    unsigned gneed[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = threadIdx.x + 256 * it;
        const int v = e >> 1, pc = e & 1;
        const int vz = v >> 6, vy = (v >> 3) & 7, vx = v & 7;
        grel[it] = ((vz * H + vy) * W + vx) * 16 + 8 * pc;
        gofs[it] = v * 16 + 8 * pc;
        gneed[it] = (1u << (vz + 1)) | (1u << (6 + vy + 1)) | (1u << (16 + vx + 1));
    }
    struct Geo { long long org; unsigned mask; int b; };
    auto geometry = [&](int t, Geo& g) {
        t = min(t, t_end - 1);
        const int tx = t % tilesX; t /= tilesX;
        const int ty = t % tilesY; t /= tilesY;
        const int tz = t % tilesZ;
        const int b = t / tilesZ;
        const int z0 = tz * CL_TZ, y0 = ty * CL_TY, x0 = tx * CL_TX;
        g.org = (((long long)b * D + z0) * H + y0) * W + x0;
        g.b = b;
        auto bits = [](int c0, int extent, int n) {
            const int lo = c0 >= 1 ? 0 : 1, hi = min(n, extent - c0 + 1);
            return ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        };
        g.mask = bits(z0, D, CL_HZ) | (bits(y0, H, CL_HY) << 6) | (bits(x0, W, CL_HX) << 16);
    };
    struct Stage { unsigned short x[NSX]; uint4 g[2]; uint4 z[2]; };
    const unsigned short* Xu = reinterpret_cast<const unsigned short*>(X);
    auto load_tile = [&](const Geo& g, Stage& st) {
#pragma unroll
        for (int it = 0; it < NSX; ++it) {
            const bool in = (g.mask & xneed[it]) == xneed[it];
            st.x[it] = Xu[g.org + (in ? xrel[it] : 0)];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const bool in = (g.mask & gneed[it]) == gneed[it];
            st.g[it] = *reinterpret_cast<const uint4*>(GY + g.org * 16 + (in ? grel[it] : 0));
            st.z[it] = *reinterpret_cast<const uint4*>(Z + g.org * 16 + (in ? grel[it] : 0));
        }
    };
    auto store_tile = [&](const Geo& g, const Stage& st) {
#pragma unroll
        for (int it = 0; it < NSX; ++it) {
            if ((int)threadIdx.x + 256 * it < CL_NH) {
                const bool in = (g.mask & xneed[it]) == xneed[it];
                const unsigned short v = in ? st.x[it] : (unsigned short)0;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int e = xhx[it] - dx;
                    if (e >= 0 && e < 8) Xs[dx][xrow[it]][e] = v;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const bool in = (g.mask & gneed[it]) == gneed[it];
            const int c0 = 8 * ((threadIdx.x + 256 * it) & 1);
            Vec16<bf16> gv, zv, o;
            gv.v = st.g[it];
            zv.v = st.z[it];
            const float* tb = &tab[g.b][0][c0];
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const float z = zv.get(k2);
                float gg = gv.get(k2) * tb[2 * 16 + k2];
                if (!(tb[k2] * z + tb[16 + k2] > 0.f)) gg = 0.f;
                o.set(k2, gg);
            }
            uint4 v1 = o.v, v2 = st.z[it];
            if (!in) { v1 = make_uint4(0, 0, 0, 0); v2 = make_uint4(0, 0, 0, 0); }
            *reinterpret_cast<uint4*>(Gt1 + gofs[it]) = v1;
            *reinterpret_cast<uint4*>(Gt2 + gofs[it]) = v2;
        }
    };
    auto compute = [&](unsigned mask) {                         // wave w: z-slice w of the tile = 2 k-steps of 4 x-rows
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int row = wave * 8 + 4 * s + kg;
            const unsigned short* g1 = Gt1 + (row * 8 + q) * 16 + 4 * p;
            const unsigned short* g2 = Gt2 + (row * 8 + q) * 16 + 4 * p;
            const bf16x8 b1 = tr_frag(g1, g1 + 4 * 16);         // g'[k = voxel x][n = channel r]
            const bf16x8 b2 = tr_frag(g2, g2 + 4 * 16);         // z
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gg = (float)b1[e], zz = (float)b2[e];
                sg += gg;
                sgz += gg * zz;
                sz += zz;
            }
            // column 0 of the third product: 1 for the voxels of this row that lie inside the volume (SX counts x only under those)
            unsigned xb = (mask >> 17) & 0xffu;
            if (r != 0 || !((mask >> (wave + 1)) & (mask >> (7 + 4 * s + kg)) & 1u)) xb = 0u;
            uint4 vw;
            vw.x = ((xb & 1u) ? 0x3f80u : 0u) | ((xb & 2u) ? 0x3f800000u : 0u);
            vw.y = ((xb & 4u) ? 0x3f80u : 0u) | ((xb & 8u) ? 0x3f800000u : 0u);
            vw.z = ((xb & 16u) ? 0x3f80u : 0u) | ((xb & 32u) ? 0x3f800000u : 0u);
            vw.w = ((xb & 64u) ? 0x3f80u : 0u) | ((xb & 128u) ? 0x3f800000u : 0u);
            const bf16x8 b3 = __builtin_bit_cast(bf16x8, vw);
            const int hrow = (wave * CL_HY + 4 * s + kg) * 8;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                uint4 a = *reinterpret_cast<const uint4*>(&Xs[0][0][0] + aoff[mt] + hrow);
                if (!aok[mt]) a = make_uint4(0, 0, 0, 0);
                const bf16x8 af = __builtin_bit_cast(bf16x8, a);
                acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b1, acc1[mt], 0, 0, 0);
                acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b2, acc2[mt], 0, 0, 0);
                acc3[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b3, acc3[mt], 0, 0, 0);
            }
        }
    };
    auto flush = [&](int n) {                                   // (uniform) this workgroup's sums for sample n -> its row
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 16 * mt + 4 * kg + i;
                if (t < 27) {
                    red[wave][FB_GX + t * 16 + r] = acc1[mt][i];
                    red[wave][FB_ZX + t * 16 + r] = acc2[mt][i];
                    if (r == 0) red[wave][FB_SX + t] = acc3[mt][i];
                }
            }
        {
            float a = sg, b2 = sgz, c2 = sz;
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b2 += __shfl_xor(b2, 16, 64); b2 += __shfl_xor(b2, 32, 64);
            c2 += __shfl_xor(c2, 16, 64); c2 += __shfl_xor(c2, 32, 64);
            if (kg == 0) { red[wave][FB_SG + r] = a; red[wave][FB_SGZ + r] = b2; red[wave][FB_SZ + r] = c2; }
        }
        __syncthreads();
        for (int o = threadIdx.x; o < FB_ROW; o += 256) {
            if (o >= FB_SX + 27 && o < FB_SG) continue;         // padding
            part[((long long)n * gridDim.x + blockIdx.x) * FB_ROW + o] = red[0][o] + red[1][o] + red[2][o] + red[3][o];
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc1[mt] = acc2[mt] = acc3[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        sg = sgz = sz = 0.f;
    };

    Stage st[WC1_DEPTH];
    Geo gg[WC1_DEPTH];
    int acc_b = -1;
#pragma unroll
    for (int d = 0; d < WC1_DEPTH; ++d) { geometry(tile + d * t_stride, gg[d]); load_tile(gg[d], st[d]); }
    for (; tile < t_end; tile += WC1_DEPTH * t_stride) {
#pragma unroll
        for (int d = 0; d < WC1_DEPTH; ++d) {
            if (tile + d * t_stride < t_end) {                  // (uniform)
                const unsigned cur_mask = gg[d].mask;
                const int cur_b = gg[d].b;
                if (cur_b != acc_b) { if (acc_b >= 0) flush(acc_b); acc_b = cur_b; }
                __syncthreads();
                store_tile(gg[d], st[d]);
                geometry(tile + (d + WC1_DEPTH) * t_stride, gg[d]);
                load_tile(gg[d], st[d]);
                __syncthreads();
                compute(cur_mask);
            }
        }
    }
    if (acc_b >= 0) flush(acc_b);
}

// sum of the per-workgroup rows of one sample: tot[n][o] (double, zeroed by first_block_bwd_kernel).  blockIdx.z splits the rows
// (a 16-workgroup reduce over the 7.7 MB of rows took 128 us); the slices meet in double atomics (order-dependent only in the last bits)
__global__ __launch_bounds__(256) void first_block_reduce_kernel(const float* __restrict__ part, int rows, double* __restrict__ tot) {
    const int n = blockIdx.y, o = blockIdx.x * 256 + threadIdx.x;
    if (o >= FB_ROW) return;
    const int per = (rows + gridDim.z - 1) / gridDim.z, w0 = blockIdx.z * per, w1 = min(rows, w0 + per);
    double s = 0.0;
    const float* p = part + (long long)n * rows * FB_ROW + o;
    for (int w = w0; w < w1; ++w) s += (double)p[(long long)w * FB_ROW];
    if (w1 > w0) atomicAdd(&tot[(long long)n * FB_ROW + o], s);
}

// one workgroup: group sums -> {alpha, beta, gamma} per (sample, channel) -> dW, db (convolution), dgamma, dbeta (normalisation)
__global__ __launch_bounds__(256) void first_block_finalize_kernel(const double* __restrict__ tot, int B, long long V, FbFwd f,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   float* __restrict__ dw, float* __restrict__ dbias, long long s_t,
                                                                   long long s_c, long long s_n) {
    __shared__ double al[16][16], be[16][16], ga[16][16];       // [sample][channel]
    const int G = f.G, cpg = 16 / G;
    if (threadIdx.x < B * 16) {
        const int n = threadIdx.x >> 4, c = threadIdx.x & 15, g = c / cpg, sn = f.Nb == 1 ? 0 : n;
        const double mu = f.stats[((long long)sn * G + g) * 2], rs = f.stats[((long long)sn * G + g) * 2 + 1];
        double A = 0.0, Bv = 0.0;                              // sums over the group (and, BatchNorm, over the samples)
        for (int m = (f.Nb == 1 ? 0 : n); m < (f.Nb == 1 ? B : n + 1); ++m)
            for (int j = 0; j < cpg; ++j) {
                const int cj = g * cpg + j;
                const double gm = f.gamma ? (double)f.gamma[cj] : 1.0;
                const double SG = tot[(long long)m * FB_ROW + FB_SG + cj], SGZ = tot[(long long)m * FB_ROW + FB_SGZ + cj];
                A += gm * SG;
                Bv += gm * rs * (SGZ - mu * SG);
            }
        const double cnt = (double)V * cpg * (f.Nb == 1 ? B : 1), pa = A / cnt, pb = Bv / cnt;
        const double gm = f.gamma ? (double)f.gamma[c] : 1.0;
        al[n][c] = rs * gm;
        be[n][c] = -rs * rs * pb;
        ga[n][c] = rs * (rs * pb * mu - pa);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 27 * 16; o += 256) {
        const int t = o >> 4, c = o & 15;
        double s = 0.0;
        for (int n = 0; n < B; ++n) {
            const double* T = tot + (long long)n * FB_ROW;
            s += al[n][c] * T[FB_GX + o] + be[n][c] * T[FB_ZX + o] + ga[n][c] * T[FB_SX + t];
        }
        dw[t * s_t + c * s_n] = (float)s;                       // (one input channel: s_c unused)
    }
    if (threadIdx.x < 16) {
        const int c = threadIdx.x, g = c / cpg;
        double sb = 0.0, dg = 0.0, db = 0.0;
        for (int n = 0; n < B; ++n) {
            const int sn = f.Nb == 1 ? 0 : n;
            const double mu = f.stats[((long long)sn * G + g) * 2], rs = f.stats[((long long)sn * G + g) * 2 + 1];
            const double* T = tot + (long long)n * FB_ROW;
            sb += al[n][c] * T[FB_SG + c] + be[n][c] * T[FB_SZ + c] + ga[n][c] * (double)V;
            dg += rs * (T[FB_SGZ + c] - mu * T[FB_SG + c]);
            db += T[FB_SG + c];
        }
        if (dbias) dbias[c] = (float)sb;
        if (dgamma) dgamma[c] = (float)dg;
        if (dbeta) dbeta[c] = (float)db;
    }
}

// ------------------------------------------------------------------------------------------------
// k=2 / stride-2 weight gradient on the bf16 matrix cores (down-convolutions and, with the operand roles swapped,
// transposed convolutions):   dW[t][c_hi][c_lo] = sum_m  HI[2m + t][c_hi] * LO[m][c_lo],   t = 2x2x2 taps.
// Same structure as wgrad_k3_bf16_kernel: natural-order LDS tiles, transposed ds_read_b64_tr_b16 fragments;
// a 4x4x8 tile of LO voxels meets its 8x8x16 block of HI voxels (no halo); the four waves own two taps each.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256, 2) void wgrad_k2s2_bf16_kernel(const bf16* __restrict__ HI, const bf16* __restrict__ LO,
                                                                 float* __restrict__ part, float* __restrict__ bias_part, int B,
                                                                 int d, int h, int w, int Chi, int Clo, int nCoBlk, int nTiles,
                                                                 int tilesZ, int tilesY, int tilesX) {
    constexpr int CB = 16 * NT;
    constexpr int HZ = 2 * WG_TZ, HY = 2 * WG_TY, HX = 2 * WG_TX;     // 8 x 8 x 16 HI voxels
    constexpr int NHI = HZ * HY * HX;                                   // 1024
    // bank-conflict-free LDS images for the transposed reads, as in wgrad_k3_bf16_kernel.  HI: a lane group reads voxels 2q (+dx) of
    // the HI rows 2 kg (+dy) -- 16 q dwords apart, so the rows of kg and kg + 1 must land 8 (mod 16) dwords apart: x-row pitch
    // 16 voxels + 16 bytes (two rows = 264 dwords = 8 mod 64)
    constexpr int HXP = HX * 16 + (WG_PAD ? 8 : 0);                     // elements
    constexpr int GVS = WG_PAD && NT == 4 ? 80 : CB;
    constexpr int GRS = 8 * GVS + (!WG_PAD ? 0 : NT == 2 ? 16 : 64);
    __shared__ __attribute__((aligned(16))) unsigned short Xh[HZ * HY * HXP];
    __shared__ __attribute__((aligned(16))) unsigned short Gt[(WG_NV / 8) * GRS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kg = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int ci0 = (blockIdx.x / nCoBlk) * 16;
    const int co0 = (blockIdx.x % nCoBlk) * CB;
    const int D = 2 * d, H = 2 * h, W = 2 * w;
    f32x4 acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bsum[j] = 0.f;
    const bool do_bias = bias_part != nullptr && ci0 == 0 && wave == 0;

    const int t_per = (nTiles + (int)gridDim.y - 1) / (int)gridDim.y;      // one contiguous tile range per split (L2 locality)
    const int t_beg = blockIdx.y * t_per, t_end = min(nTiles, t_beg + t_per);
    for (int tile = t_beg; tile < t_end; ++tile) {
        int rr = tile;
        const int tx = rr % tilesX; rr /= tilesX;
        const int ty = rr % tilesY; rr /= tilesY;
        const int tz = rr % tilesZ;
        const int b = rr / tilesZ;
        const int z0 = tz * WG_TZ, y0 = ty * WG_TY, x0 = tx * WG_TX;   // LO coordinates
        __syncthreads();
        {   // ---- HI block: 1024 voxels x 32 B, all loads in flight first
            uint4 sx[NHI * 2 / 256];
#pragma unroll
            for (int it = 0; it < NHI * 2 / 256; ++it) {
                const int e = threadIdx.x + 256 * it;
                const int hv = e >> 1, half = e & 1;
                const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
                const int z = 2 * z0 + hz, y = 2 * y0 + hy, x = 2 * x0 + hx;
                sx[it] = make_uint4(0, 0, 0, 0);
                if (z < D && y < H && x < W)
                    sx[it] = *reinterpret_cast<const uint4*>(HI + ((((long long)b * D + z) * H + y) * W + x) * Chi + ci0 + 8 * half);
            }
#pragma unroll
            for (int it = 0; it < NHI * 2 / 256; ++it) {
                const int e = threadIdx.x + 256 * it;
                const int hv = e >> 1;
                *reinterpret_cast<uint4*>(Xh + (hv / HX) * HXP + (hv % HX) * 16 + 8 * (e & 1)) = sx[it];
            }
        }
        {   // ---- LO tile: 128 voxels x CB channels
            constexpr int NSG = WG_NV * (CB / 8) / 256 > 0 ? WG_NV * (CB / 8) / 256 : 1;
            uint4 sg[NSG];
#pragma unroll
            for (int it = 0; it < NSG; ++it) {
                const int e = threadIdx.x + 256 * it;
                const int v8 = e % (CB / 8), vv = e / (CB / 8);
                const int vx = vv % WG_TX, vy = (vv / WG_TX) % WG_TY, vz = vv / (WG_TX * WG_TY);
                const int z = z0 + vz, y = y0 + vy, x = x0 + vx;
                sg[it] = make_uint4(0, 0, 0, 0);
                if (e < WG_NV * (CB / 8) && z < d && y < h && x < w && co0 + 8 * v8 < Clo)
                    sg[it] = *reinterpret_cast<const uint4*>(LO + ((((long long)b * d + z) * h + y) * w + x) * Clo + co0 + 8 * v8);
            }
#pragma unroll
            for (int it = 0; it < NSG; ++it) {
                const int e = threadIdx.x + 256 * it;
                const int vv = e / (CB / 8);
                if (e < WG_NV * (CB / 8)) *reinterpret_cast<uint4*>(Gt + (vv >> 3) * GRS + (vv & 7) * GVS + 8 * (e % (CB / 8))) = sg[it];
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = 4 * s + kg, z = row >> 2, y = row & 3;
            bf16x8 bfr[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const unsigned short* g0 = Gt + row * GRS + q * GVS + 16 * j + 4 * p;
                bfr[j] = tr_frag(g0, g0 + 4 * GVS);
                if (do_bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[j] += (float)bfr[j][e];
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int t = 2 * wave + a;
                const int dz = t >> 2, dy = (t >> 1) & 1, dx = t & 1;
                // HI voxel of LO voxel (z, y, x = q (+4)) : (2z+dz, 2y+dy, 2x+dx)
                const unsigned short* a0 = Xh + ((2 * z + dz) * HY + (2 * y + dy)) * HXP + (2 * q + dx) * 16 + 4 * p;
                const bf16x8 afr = tr_frag(a0, a0 + 8 * 16);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[j], acc[a][j], 0, 0, 0);
            }
        }
    }
    const int col = lane & 15;
    float* dst = part + (long long)blockIdx.y * 8 * Chi * Clo;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int t = 2 * wave + a;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = co0 + 16 * j + col;
            if (co >= Clo) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[((long long)t * Chi + ci0 + 4 * kg + i) * Clo + co] = acc[a][j][i];
        }
    }
    if (do_bias) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float v = bsum[j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int co = co0 + 16 * j + col;
            if (kg == 0 && co < Clo) bias_part[(long long)blockIdx.y * Clo + co] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 1x1 weight gradient on the bf16 matrix cores (the projection head, UNet3D_contrastive.py:262,265: 256 -> 512 -> 256 at 12^3):
//   dW[ci][co] = sum_rows X[row][ci] * G[row][co].
// Same recipe as the k=3 / k=2 kernels: 128-row tiles of X (64 input channels: 16 per wave) and G (16*NT output channels) staged in
// natural [row][channel] order, fragments read back transposed (K = rows).  Until round 2 these two layers ran on the f32-input
// MFMA fall-back (103 us each, the longest launches of the weight-gradient stream).
// ------------------------------------------------------------------------------------------------
constexpr int W1_ROWS = 128, W1_CI = 64;
template <int NT>
__global__ __launch_bounds__(256, 2) void wgrad_1x1_bf16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ GY,
                                                                float* __restrict__ part, float* __restrict__ bias_part, long long M,
                                                                int Cin, int Cout, int nCoBlk, int nTiles) {
    constexpr int CB = 16 * NT;
    __shared__ __attribute__((aligned(16))) unsigned short Xt[W1_ROWS * W1_CI];
    __shared__ __attribute__((aligned(16))) unsigned short Gt[W1_ROWS * CB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kg = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int ci0 = (blockIdx.x / nCoBlk) * W1_CI;
    const int co0 = (blockIdx.x % nCoBlk) * CB;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bsum[j] = 0.f;
    const bool do_bias = bias_part != nullptr && ci0 == 0 && wave == 0;
    constexpr int NSX = W1_ROWS * (W1_CI / 8) / 256;                  // 4 pieces of 16 B per thread
    constexpr int NSG = W1_ROWS * (CB / 8) / 256 > 0 ? W1_ROWS * (CB / 8) / 256 : 1;
    uint4 sx[NSX], sg[NSG];
    auto load_tile = [&](int tile) {
        const long long r0 = (long long)tile * W1_ROWS;
#pragma unroll
        for (int it = 0; it < NSX; ++it) {
            const int e = threadIdx.x + 256 * it;
            const int c8 = e % (W1_CI / 8), rr = e / (W1_CI / 8);
            sx[it] = make_uint4(0, 0, 0, 0);
            if (r0 + rr < M) sx[it] = *reinterpret_cast<const uint4*>(X + (r0 + rr) * Cin + ci0 + 8 * c8);
        }
#pragma unroll
        for (int it = 0; it < NSG; ++it) {
            const int e = threadIdx.x + 256 * it;
            const int c8 = e % (CB / 8), rr = e / (CB / 8);
            sg[it] = make_uint4(0, 0, 0, 0);
            if (e < W1_ROWS * (CB / 8) && r0 + rr < M && co0 + 8 * c8 < Cout) sg[it] = *reinterpret_cast<const uint4*>(GY + (r0 + rr) * Cout + co0 + 8 * c8);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < NSX; ++it) *reinterpret_cast<uint4*>(Xt + (threadIdx.x + 256 * it) * 8) = sx[it];
#pragma unroll
        for (int it = 0; it < NSG; ++it) {
            const int e = threadIdx.x + 256 * it;
            if (e < W1_ROWS * (CB / 8)) *reinterpret_cast<uint4*>(Gt + e * 8) = sg[it];
        }
    };
    const int t_per = (nTiles + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t_beg = blockIdx.y * t_per, t_end = min(nTiles, t_beg + t_per);
    if (t_beg < t_end) load_tile(t_beg);
    for (int tile = t_beg; tile < t_end; ++tile) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (tile + 1 < t_end) load_tile(tile + 1);
#pragma unroll
        for (int s_ = 0; s_ < W1_ROWS / 32; ++s_) {
            const int v = 32 * s_ + 8 * kg + q;                         // this lane's row of the k-step (and v + 4)
            const unsigned short* a0 = Xt + v * W1_CI + 16 * wave + 4 * p;
            const bf16x8 afr = tr_frag(a0, a0 + 4 * W1_CI);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const unsigned short* g0 = Gt + v * CB + 16 * j + 4 * p;
                const bf16x8 bfr = tr_frag(g0, g0 + 4 * CB);
                if (do_bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[j] += (float)bfr[e];
                }
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, acc[j], 0, 0, 0);
            }
        }
    }
    const int col = lane & 15;
    float* dst = part + (long long)blockIdx.y * Cin * Cout;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = co0 + 16 * j + col;
        if (co >= Cout) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[((long long)ci0 + 16 * wave + 4 * kg + i) * Cout + co] = acc[j][i];
    }
    if (do_bias) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float v = bsum[j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int co = co0 + 16 * j + col;
            if (kg == 0 && co < Cout) bias_part[(long long)blockIdx.y * Cout + co] = v;
        }
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
    // four interleaved voxel streams: four independent load->fma chains in flight per thread
    float acc4[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long m0 = m_beg; m0 < m_end; m0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long m = m0 + u;
            if (m >= m_end) continue;
            long long q = m;
            const int xo = (int)(q % Wo); q /= Wo;
            const int yo = (int)(q % Ho); q /= Ho;
            const int zo = (int)(q % Do);
            const int bo = (int)(q / Do);
            int zi, yi, xi;
            if (src_voxel<MODE>(t, zo, yo, xo, Di, Hi, Wi, zi, yi, xi))
                acc4[u] += ldf(X + ((((long long)bo * Di + zi) * Hi + yi) * Wi + xi) * Cin + ci) * ldf(GY + m * Cout + co);
        }
    }
    const float acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    part[(long long)blockIdx.x * L + j] = acc;
}

// 1x1 weight gradient for the 2-class heads (Cin = 16, Cout = 2): every thread keeps all CIN*COUT partial
// products in registers over a strided voxel stream (one 32-B + one 8-B load per voxel), block-reduces them
// through LDS and writes one partial row.  Pure HBM stream.
template <typename TX, typename TG, int CIN, int COUT>
__global__ __launch_bounds__(256) void wgrad_1x1_skinny_kernel(const TX* __restrict__ X, const TG* __restrict__ GY,
                                                               float* __restrict__ part, long long M) {
    __shared__ float red[4][CIN * COUT];
    float acc[CIN * COUT];
#pragma unroll
    for (int k = 0; k < CIN * COUT; ++k) acc[k] = 0.f;
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
        float xv[CIN], gv[COUT];
        constexpr int VN = Vec16<TX>::N;
#pragma unroll
        for (int c0 = 0; c0 < CIN; c0 += VN) {
            const Vec16<TX> v = ld16(X + m * CIN + c0);
#pragma unroll
            for (int k = 0; k < VN; ++k) xv[c0 + k] = v.get(k);
        }
#pragma unroll
        for (int k = 0; k < COUT; ++k) gv[k] = ldf(GY + m * COUT + k);
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int k = 0; k < COUT; ++k) acc[c * COUT + k] += xv[c] * gv[k];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < CIN * COUT; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) red[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < CIN * COUT)   // partial layout [t=0][ci][co]
        part[(long long)blockIdx.x * CIN * COUT + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// y[m, 0..COUT) = bias + sum_c x[m, c] * W[c][.]   (Cin = 16 -> 2 logits: one 32-B load per voxel)
template <typename TI, int CIN, int COUT>
__global__ __launch_bounds__(256) void head_1x1_fwd_kernel(const TI* __restrict__ X, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ Y, long long M) {
    __shared__ float w[CIN * COUT + COUT];
    if (threadIdx.x < CIN * COUT) w[threadIdx.x] = W[threadIdx.x];
    if (threadIdx.x < COUT) w[CIN * COUT + threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
    __syncthreads();
    constexpr int VN = Vec16<TI>::N;
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
        float o[COUT];
#pragma unroll
        for (int k = 0; k < COUT; ++k) o[k] = w[CIN * COUT + k];
#pragma unroll
        for (int c0 = 0; c0 < CIN; c0 += VN) {
            const Vec16<TI> v = ld16(X + m * CIN + c0);
#pragma unroll
            for (int e = 0; e < VN; ++e)
#pragma unroll
                for (int k = 0; k < COUT; ++k) o[k] += v.get(e) * w[(c0 + e) * COUT + k];
        }
#pragma unroll
        for (int k = 0; k < COUT; ++k) Y[m * COUT + k] = o[k];
    }
}

// gx[m, 0..COUT) = sum_k g[m, k] * W[k][.]   (2 logit gradients -> 16 channels, 16-B stores)
template <typename TO, int CIN, int COUT>
__global__ __launch_bounds__(256) void head_1x1_bwd_kernel(const float* __restrict__ G, const float* __restrict__ W,
                                                           TO* __restrict__ Y, long long M, int accumulate) {
    __shared__ float w[CIN * COUT];
    if (threadIdx.x < CIN * COUT) w[threadIdx.x] = W[threadIdx.x];
    __syncthreads();
    constexpr int VN = Vec16<TO>::N;
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
        float g[CIN];
#pragma unroll
        for (int k = 0; k < CIN; ++k) g[k] = G[m * CIN + k];
#pragma unroll
        for (int c0 = 0; c0 < COUT; c0 += VN) {
            Vec16<TO> o;
            if (accumulate) o = ld16(Y + m * COUT + c0);
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                float v = accumulate ? o.get(e) : 0.f;
#pragma unroll
                for (int k = 0; k < CIN; ++k) v += g[k] * w[k * COUT + c0 + e];
                o.set(e, v);
            }
            st16(Y + m * COUT + c0, o);
        }
    }
}

// ordered sum over the P partial slabs of a weight-gradient launch:  out[t*s_t + ci*s_c + co*s_n] = sum_p part[p][(t, ci, co)].
// One launch serves up to two jobs (the weight slabs and the bias-gradient partials of the same wgrad call): blocks
// [0, nb0) belong to job 0, the rest to job 1.
struct ReduceJob { const float* part; float* out; int P, L, Cin, Cout; long long s_t, s_c, s_n; };
// VEC = 4: a thread owns 4 consecutive outputs (one 16-byte load per partial row: the slabs are the bulk of a weight-gradient call's
// HBM traffic, ~25 MB per call); 256 threads = 32 output quads x 8 partial lanes.  VEC = 1: any L.
template <int VEC>
__global__ __launch_bounds__(256) void reduce_partials_kernel(ReduceJob j0, ReduceJob j1, int nb0) {
    __shared__ float sm[8][32 * VEC + 1];
    const bool second = (int)blockIdx.x >= nb0;
    const ReduceJob& jb = second ? j1 : j0;
    const int bid = second ? blockIdx.x - nb0 : blockIdx.x;
    const int o = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int i = (bid * 32 + o) * VEC;
    const int L = jb.L, P = jb.P;
    float s[4][VEC];                                       // four independent chains: the loads of a thread are 8*L floats apart
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) s[c][v] = 0.f;
    auto add = [&](int c, long long p) {
        if (VEC == 4) {
            const float4 q = *reinterpret_cast<const float4*>(jb.part + p * L + i);
            s[c][0] += q.x; s[c][1 % VEC] += q.y; s[c][2 % VEC] += q.z; s[c][3 % VEC] += q.w;
        } else {
            s[c][0] += jb.part[p * L + i];
        }
    };
    if (i < L) {
        int p = pl;
        for (; p + 24 < P; p += 32) { add(0, p); add(1, p + 8); add(2, p + 16); add(3, p + 24); }
        for (; p < P; p += 8) add(0, p);
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) sm[pl][o * VEC + v] = (s[0][v] + s[1][v]) + (s[2][v] + s[3][v]);
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * VEC; e += 256) {
        const int ii = bid * 32 * VEC + e;
        if (ii >= L) continue;
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += sm[k][e];
        const int co = ii % jb.Cout, ci = (ii / jb.Cout) % jb.Cin, t = ii / (jb.Cout * jb.Cin);
        jb.out[t * jb.s_t + ci * jb.s_c + co * jb.s_n] = tot;
    }
}

// column sums of an (rows, C) matrix, stage 1: partial[blk][c].  16-byte loads, channel group fixed per thread
// (threads of a block tile [rows-per-iteration][C/VN]), partials combined through LDS.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, float* __restrict__ part, long long rows, int C,
                                                     long long rows_per_block) {
    constexpr int VN = Vec16<T>::N;
    __shared__ float sm[256][VN + 1];
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    if (C <= 8 && 256 % C == 0) {          // skinny widths (2-class logits): threads tile [rows][C], LDS tree over rows
        const int c = threadIdx.x % C, rr = threadIdx.x / C, rpi2 = 256 / C;
        float s = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;     // four loads in flight per thread (a single dependent chain was latency-bound)
        long long r = r0 + rr;
        for (; r + 3 * rpi2 < r1; r += 4 * rpi2) {
            s += ldf(X + r * C + c); s1 += ldf(X + (r + rpi2) * C + c); s2 += ldf(X + (r + 2 * rpi2) * C + c); s3 += ldf(X + (r + 3 * rpi2) * C + c);
        }
        for (; r < r1; r += rpi2) s += ldf(X + r * C + c);
        s = (s + s1) + (s2 + s3);
        sm[threadIdx.x][0] = s;
        __syncthreads();
        if (threadIdx.x < C) {
            float tot = 0.f;
            for (int q = 0; q < rpi2; ++q) tot += sm[q * C + threadIdx.x][0];
            part[(long long)blockIdx.x * C + threadIdx.x] = tot;
        }
        return;
    }
    if (C % VN != 0 || C / VN > 256) {     // other odd widths: scalar path
        for (int c = threadIdx.x; c < C; c += 256) {
            float s = 0.f;
            for (long long r = r0; r < r1; ++r) s += ldf(X + r * C + c);
            part[(long long)blockIdx.x * C + c] = s;
        }
        return;
    }
    const int ngrp = C / VN, rpi = 256 / ngrp;
    const int cg = threadIdx.x % ngrp, rr = threadIdx.x / ngrp;
    float a[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) a[k] = 0.f;
    if (rr < rpi)
        for (long long r = r0 + rr; r < r1; r += rpi) {
            const Vec16<T> v = ld16(X + r * C + cg * VN);
#pragma unroll
            for (int k = 0; k < VN; ++k) a[k] += v.get(k);
        }
#pragma unroll
    for (int k = 0; k < VN; ++k) sm[threadIdx.x][k] = a[k];
    __syncthreads();
    for (int o = threadIdx.x; o < C; o += 256) {
        const int og = o / VN, ok = o % VN;
        float s = 0.f;
        for (int q = 0; q < rpi; ++q) s += sm[q * ngrp + og][ok];
        part[(long long)blockIdx.x * C + o] = s;
    }
}

// small-L reduction: one block per output, 256 lanes over the partials (column sums: L = C <= 512, P up to 2048)
__global__ __launch_bounds__(256) void reduce_partials_small_kernel(const float* __restrict__ part, int P, int L, float* __restrict__ out,
                                                                    int Cin, int Cout, long long s_t, long long s_c, long long s_n) {
    __shared__ float red[17];
    const int i = blockIdx.x;
    float s = 0.f;
    for (int p = threadIdx.x; p < P; p += 256) s += part[(long long)p * L + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const int co = i % Cout, ci = (i / Cout) % Cin, t = i / (Cout * Cin);
        out[t * s_t + ci * s_c + co * s_n] = s;
    }
}

static void launch_reduce_partials(const float* part, int P, int L, float* out, int Cin, int Cout, long long s_t, long long s_c,
                                   long long s_n, dycon_stream_t stream, const float* bias_part = nullptr, float* bias_out = nullptr) {
    if (!bias_part && L <= 512 && P > 64) {
        reduce_partials_small_kernel<<<L, 256, 0, stream>>>(part, P, L, out, Cin, Cout, s_t, s_c, s_n);
        return;
    }
    const ReduceJob j0{part, out, P, L, Cin, Cout, s_t, s_c, s_n};
    const ReduceJob j1{bias_part, bias_out, P, bias_part ? Cout : 0, 1, Cout, 0, 0, 1};
    const bool vec = L % 4 == 0 && (!bias_part || Cout % 4 == 0) && ((uintptr_t)part & 15) == 0 && (!bias_part || ((uintptr_t)bias_part & 15) == 0);
    if (vec) {
        const int nb0 = cdiv(L, 128), nb1 = bias_part ? cdiv(Cout, 128) : 0;
        reduce_partials_kernel<4><<<nb0 + nb1, 256, 0, stream>>>(j0, j1, nb0);
    } else {
        const int nb0 = cdiv(L, 32), nb1 = bias_part ? cdiv(Cout, 32) : 0;
        reduce_partials_kernel<1><<<nb0 + nb1, 256, 0, stream>>>(j0, j1, nb0);
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

// split-K plan: small spatial levels have too few 64-row blocks to fill 256 CUs and a long serial K loop
struct SplitK { int splits, kc_per_split; };
static SplitK splitk_plan(int dtype, int mode, int scatter, long long M, int N, int Cin) {
    const int KC = dtype == DYCON_BF16 ? 32 : 16;
    const int Tn = mode == DYCON_CONV_K3 ? 27 : mode == DYCON_CONV_K2S2 ? 8 : 1;
    const int nKC = (Tn * Cin + KC - 1) / KC;
    const long long wgs = ((M + 63) / 64) * (((N + 15) / 16 + NTB - 1) / NTB);
    SplitK p{1, nKC};
    if (scatter || wgs >= 512 || nKC < 16) return p;
    long long s = (1024 + wgs - 1) / wgs;
    if (s > nKC / 6) s = nKC / 6;
    if (s < 2) return p;
    p.kc_per_split = (int)((nKC + s - 1) / s);
    p.splits = (nKC + p.kc_per_split - 1) / p.kc_per_split;
    return p;
}

// LDS-tiled kernel for the small levels: bf16, k3, channel counts that fill its 32-wide k-steps and 128-wide column tiles
static bool conv_tile_ok(int dtype, int mode, int scatter, long long DHW, int Cin, int N) {
    return dtype == DYCON_BF16 && mode == DYCON_CONV_K3 && !scatter && Cin % 32 == 0 && N % CT_BN == 0 && Cin >= 64 && DHW < 13824;
}
// tunables read once from the environment (diagnostic sweeps only; the defaults are what ships)
static long long env_ll(const char* name, long long dflt) {
    const char* v = getenv(name);
    return v && *v ? atoll(v) : dflt;
}
static SplitK conv_tile_plan(long long M, int N, int Cin) {
    static const long long target_wgs = env_ll("DYCON_TILE_SPLIT_WGS", 512);
    const int nKC = 27 * Cin / 32;
    const long long wgs = ((M + CT_BM - 1) / CT_BM) * (N / CT_BN);
    SplitK p{1, nKC};
    if (wgs >= 384) return p;
    long long s = target_wgs / wgs;                           // 512 workgroups (2 of the 4 slots per CU, one round): 5.66-5.68 ms/step against 5.74 at 768 and 5.72 at 256
    if (s > nKC / 8) s = nKC / 8;                             // at least 8 k-steps per split
    if (s < 2) return p;
    p.kc_per_split = (int)((nKC + s - 1) / s);
    if (Cin % 64 == 0) p.kc_per_split = (p.kc_per_split + 1) & ~1;    // whole 64-channel k-steps (conv_k3_tile_kernel<2>)
    p.splits = (nKC + p.kc_per_split - 1) / p.kc_per_split;
    return p;
}

extern "C" int dycon_pack_batch(const dycon_pack_job_t* jobs_dev, int njobs, int blocks_per_job, dycon_stream_t stream) {
    DYCON_REQUIRE(jobs_dev && njobs > 0 && njobs <= 65535 && blocks_per_job > 0, "pack_batch: bad arguments");
    pack_batch_kernel<<<dim3(blocks_per_job, njobs), 256, 0, stream>>>(jobs_dev);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

template <typename T, int MODE, bool SC>
static void launch_gemm(const void* x, const void* wf, const float* bias, void* y, int accumulate, int B, int Di, int Hi,
                        int Wi, int Cin, int N, int Cout, float* workspace, int defer_finish, dycon_stream_t stream) {
    int Do, Ho, Wo;
    row_grid(MODE, Di, Hi, Wi, Do, Ho, Wo);
    const long long M = (long long)B * Do * Ho * Wo;
    const int NT = (N + 15) / 16;
    const int Tn = MODE == DYCON_CONV_K3 ? 27 : MODE == DYCON_CONV_K2S2 ? 8 : 1;
    const int nKC = (Tn * Cin + Frag<T>::KC - 1) / Frag<T>::KC;
    const SplitK sk = splitk_plan(sizeof(T) == 2 ? DYCON_BF16 : DYCON_F32, MODE, SC, M, N, Cin);
    const bool split = sk.splits > 1 && workspace != nullptr;
    dim3 grid(cdiv(M, 64), cdiv(NT, NTB), split ? sk.splits : 1);
    conv_gemm_kernel<T, MODE, SC><<<grid, 256, 0, stream>>>((const T*)x, (const T*)wf, bias, (T*)y, B, Di, Hi, Wi, Cin, Do, Ho,
                                                            Wo, N, Cout, NT, nKC, accumulate, split ? workspace : nullptr,
                                                            sk.kc_per_split, split && defer_finish && N % 8 == 0);
    if (split && !defer_finish) {
        long long blocks = (M * N / 4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        splitk_finish_kernel<T><<<(int)blocks, 256, 0, stream>>>(workspace, sk.splits, M * N, N, bias, (T*)y, accumulate);
    }
}

extern "C" size_t dycon_conv_gemm_workspace(int dtype, int mode, int scatter, int B, int Di, int Hi, int Wi, int Cin, int N) {
    int Do, Ho, Wo;
    row_grid(mode, Di, Hi, Wi, Do, Ho, Wo);
    const long long M = (long long)B * Do * Ho * Wo;
    const SplitK sk = conv_tile_ok(dtype, mode, scatter, (long long)Di * Hi * Wi, Cin, N) ? conv_tile_plan(M, N, Cin)
                                                                                       : splitk_plan(dtype, mode, scatter, M, N, Cin);
    return sk.splits > 1 ? (size_t)sk.splits * M * N * sizeof(float) : 0;
}

extern "C" int dycon_conv_gemm_splits(int dtype, int mode, int scatter, int B, int Di, int Hi, int Wi, int Cin, int N) {
    int Do, Ho, Wo;
    row_grid(mode, Di, Hi, Wi, Do, Ho, Wo);
    const long long M = (long long)B * Do * Ho * Wo;
    const SplitK sk = conv_tile_ok(dtype, mode, scatter, (long long)Di * Hi * Wi, Cin, N) ? conv_tile_plan(M, N, Cin)
                                                                                       : splitk_plan(dtype, mode, scatter, M, N, Cin);
    return sk.splits;
}

// defer_finish = 1 (split-K shapes with a workspace only): leave the partial slabs in `workspace` and do NOT launch the finish --
// the caller completes the convolution with dycon_norm_fwd_slab (bias, ordered sum, rounding, norm in one launch).
extern "C" int dycon_conv_gemm_ex(const void* x, const void* wfrag, const float* bias, void* y, int dtype, int mode,
                                  int scatter, int accumulate, int B, int Di, int Hi, int Wi, int Cin, int N, int Cout,
                                  float* workspace, size_t ws_bytes, int defer_finish, dycon_stream_t stream);

extern "C" int dycon_conv_gemm(const void* x, const void* wfrag, const float* bias, void* y, int dtype, int mode,
                               int scatter, int accumulate, int B, int Di, int Hi, int Wi, int Cin, int N, int Cout,
                               float* workspace, size_t ws_bytes, dycon_stream_t stream) {
    return dycon_conv_gemm_ex(x, wfrag, bias, y, dtype, mode, scatter, accumulate, B, Di, Hi, Wi, Cin, N, Cout, workspace, ws_bytes, 0,
                              stream);
}

static thread_local float* g_stat_part = nullptr;      // set by dycon_conv_gemm_stats around its call of dycon_conv_gemm_ex

static bool conv_p32_shape(int dtype, int mode, int scatter, int accumulate, int B, int Di, int Hi, int Wi, int Cin, int Cout) {
    static const bool p32_on = env_ll("DYCON_P32", 1) != 0;
    const int nTiles = B * cdiv(Di, CL_TZ) * cdiv(Hi, CL_TY) * cdiv(Wi, CL_TX);
    return p32_on && dtype == DYCON_BF16 && mode == DYCON_CONV_K3 && !scatter && !accumulate && Cin == 32 && Cout == 32 &&
           (long long)Di * Hi * Wi >= 13824 && nTiles >= 1024;
}
// rows per sample of the statistics partials a dycon_conv_gemm_stats call of this shape writes (0: shape not served)
extern "C" int dycon_conv_stats_chunks(int dtype, int mode, int B, int Di, int Hi, int Wi, int Cin, int Cout) {
    const int nTiles = B * cdiv(Di, CL_TZ) * cdiv(Hi, CL_TY) * cdiv(Wi, CL_TX);
    // the persistent kernels of the 16-channel level (conv_k3_p16 16 -> 16, conv_k3_c1 1 -> 16): one row per workgroup of their grids
    if (dtype == DYCON_BF16 && mode == DYCON_CONV_K3 && Cout == 16 && (Cin == 16 || Cin == 1) && (long long)Di * Hi * Wi >= 13824)
        return 8 * min(cdiv(nTiles, 8), Cin == 16 ? 32 * P16_WGS : 128);
    if (!conv_p32_shape(dtype, mode, 0, 0, B, Di, Hi, Wi, Cin, Cout)) return 0;
    return 8 * min(cdiv(nTiles, 8), 32);
}
extern "C" int dycon_conv_gemm_ex(const void* x, const void* wfrag, const float* bias, void* y, int dtype, int mode,
                                  int scatter, int accumulate, int B, int Di, int Hi, int Wi, int Cin, int N, int Cout,
                                  float* workspace, size_t ws_bytes, int defer_finish, dycon_stream_t stream);
// dycon_conv_gemm (k=3, no scatter / accumulation) on a shape dycon_conv_stats_chunks serves, which ALSO leaves per-(sample, chunk,
// channel) {sum, sum of squares} of the stored outputs in stat_part ([B][chunks][Cout][2] floats): the statistics pass of the
// normalisation that follows is then dycon_norm_fwd_parts (finalize + apply) instead of a read of the whole tensor.
extern "C" int dycon_conv_gemm_stats(const void* x, const void* wfrag, const float* bias, void* y, int dtype, int B, int Di, int Hi,
                                     int Wi, int Cin, int Cout, float* stat_part, size_t stat_bytes, dycon_stream_t stream) {
    const int chunks = dycon_conv_stats_chunks(dtype, DYCON_CONV_K3, B, Di, Hi, Wi, Cin, Cout);
    DYCON_REQUIRE(chunks > 0, "conv_gemm_stats: shape not served (ask dycon_conv_stats_chunks)");
    DYCON_REQUIRE(stat_part && stat_bytes >= (size_t)B * chunks * Cout * 2 * sizeof(float), "conv_gemm_stats: statistics buffer too small");
    g_stat_part = stat_part;
    const int rc = dycon_conv_gemm_ex(x, wfrag, bias, y, dtype, DYCON_CONV_K3, 0, 0, B, Di, Hi, Wi, Cin, Cout, Cout, nullptr, 0, 0, stream);
    g_stat_part = nullptr;
    return rc;
}

extern "C" int dycon_conv_gemm_ex(const void* x, const void* wfrag, const float* bias, void* y, int dtype, int mode,
                                  int scatter, int accumulate, int B, int Di, int Hi, int Wi, int Cin, int N, int Cout,
                                  float* workspace, size_t ws_bytes, int defer_finish, dycon_stream_t stream) {
    DYCON_REQUIRE(x && wfrag && y, "conv_gemm: null pointer");
    DYCON_REQUIRE(!defer_finish || (!accumulate && workspace && dycon_conv_gemm_splits(dtype, mode, scatter, B, Di, Hi, Wi, Cin, N) > 1 &&
                                    ws_bytes >= dycon_conv_gemm_workspace(dtype, mode, scatter, B, Di, Hi, Wi, Cin, N)),
                  "conv_gemm: defer_finish needs a split-K shape, its workspace and no accumulation");
    DYCON_REQUIRE(B > 0 && Di > 0 && Hi > 0 && Wi > 0 && Cin > 0 && N > 0 && Cout > 0, "conv_gemm: bad shape");
    DYCON_REQUIRE(mode >= 0 && mode <= 2, "conv_gemm: bad mode %d", mode);
    const bool first_layer_lds = dtype == DYCON_BF16 && mode == DYCON_CONV_K3 && !scatter && Cin == 1 &&
                                 (Cout == 16 || Cout == 32 || Cout == 64) && (long long)Di * Hi * Wi >= 13824;
    DYCON_REQUIRE(first_layer_lds || Cin % (dtype == DYCON_BF16 ? 8 : 4) == 0, "conv_gemm: Cin=%d not a multiple of the fragment width", Cin);
    DYCON_REQUIRE(N % 16 == 0, "conv_gemm: N=%d not a multiple of 16 (use dycon_conv_direct)", N);
    DYCON_REQUIRE(!scatter || (mode == DYCON_CONV_1X1 && N == 8 * Cout), "conv_gemm: scatter needs mode 1x1 and N == 8*Cout");
    DYCON_REQUIRE(scatter || N == Cout, "conv_gemm: N must equal Cout without scatter");
    DYCON_REQUIRE(mode != DYCON_CONV_K2S2 || (Di % 2 == 0 && Hi % 2 == 0 && Wi % 2 == 0), "conv_gemm: k2s2 needs even dims");
    // large spatial levels, bf16: LDS-halo kernel
    if (dtype == DYCON_BF16 && mode == DYCON_CONV_K3 && !scatter &&
        ((Cin == 1 && (Cout == 16 || Cout == 32 || Cout == 64)) ||
         ((Cin == 16 || Cin == 48 || Cin % 32 == 0) && (Cout == 16 || Cout == 32 || Cout % 64 == 0 || Cout % 48 == 0))) &&
        (long long)Di * Hi * Wi >= 13824) {
        const int tz = cdiv(Di, CL_TZ), ty = cdiv(Hi, CL_TY), tx = cdiv(Wi, CL_TX);
        const int NT = Cout / 16;
        const int ntb = Cout % 64 == 0 ? CL_NTB64 : (Cout % 48 == 0 ? 3 : NT);
        const int nTiles = B * tz * ty * tx;
        // persistent kernels of the 16-channel level: 8 XCDs x up to 64 (p16: 2 per CU) / 128 (c1) workgroups
        if ((Cin == 16 && Cout == 16) || (Cin == 1 && (Cout == 16 || Cout == 32 || Cout == 64))) {
            const int per_xcd = min(cdiv(nTiles, 8), Cin == 16 ? 32 * P16_WGS : 128);
            const bf16 *xp = (const bf16*)x, *wp = (const bf16*)wfrag;
            bf16* yp = (bf16*)y;
            if (Cin == 16 && accumulate) conv_k3_p16_kernel<P16_DEPTH, true><<<8 * per_xcd, 256, 0, stream>>>(xp, wp, bias, yp, B, Di, Hi, Wi, tz, ty, tx, nTiles);
            else if (Cin == 16 && g_stat_part) conv_k3_p16_kernel<P16_DEPTH, false, true><<<8 * per_xcd, 256, 0, stream>>>(xp, wp, bias, yp, B, Di, Hi, Wi, tz, ty, tx, nTiles, g_stat_part);
            else if (Cin == 16) conv_k3_p16_kernel<P16_DEPTH, false><<<8 * per_xcd, 256, 0, stream>>>(xp, wp, bias, yp, B, Di, Hi, Wi, tz, ty, tx, nTiles);
            else if (Cout == 16 && g_stat_part && !accumulate) conv_k3_c1_kernel<1, true><<<8 * per_xcd, 256, 0, stream>>>(xp, wp, bias, yp, B, Di, Hi, Wi, tz, ty, tx, nTiles, 0, g_stat_part);
            else if (Cout == 16) conv_k3_c1_kernel<1><<<8 * per_xcd, 256, 0, stream>>>(xp, wp, bias, yp, B, Di, Hi, Wi, tz, ty, tx, nTiles, accumulate);
            else if (Cout == 32) conv_k3_c1_kernel<2><<<8 * per_xcd, 256, 0, stream>>>(xp, wp, bias, yp, B, Di, Hi, Wi, tz, ty, tx, nTiles, accumulate);
            else conv_k3_c1_kernel<4><<<8 * per_xcd, 256, 0, stream>>>(xp, wp, bias, yp, B, Di, Hi, Wi, tz, ty, tx, nTiles, accumulate);
            DYCON_LAUNCH_CHECK();
            return DYCON_OK;
        }
        // 32 -> 32 with several tiles per CU: persistent kernel, weights stationary in LDS (one workgroup per CU)
        static const bool p32_on = env_ll("DYCON_P32", 1) != 0;
        if (p32_on && Cin == 32 && Cout == 32 && !accumulate && nTiles >= 1024) {
            const int per_xcd = min(cdiv(nTiles, 8), 32);
            static const long long p32_nw = env_ll("DYCON_P32_WAVES", 8);
            // DYCON_P32X=1: the 32x32x16 MFMA form (conv_k3_p32x_kernel).  Measured SLOWER (30.9 vs 28.5 us per launch at 48^3, step
            // unchanged): equal MFMA cycles per tap, and the chip holds a lower clock on the 32x32 shape (MI355X_MICROARCH.md, DVFS
            // give-back item 7) -- what the freed issue slots return does not make up for it.  Kept (tested), off.  Read per call:
            // the tests switch it inside one process.
            const bool p32x = env_ll("DYCON_P32X", 0) != 0;
#define DYCON_P32(KRN, NWV, STV) KRN<NWV, STV><<<8 * per_xcd, 64 * NWV, 0, stream>>>((const bf16*)x, (const bf16*)wfrag, bias, (bf16*)y, B, Di, Hi, Wi, tz, ty, tx, nTiles, g_stat_part)
            if (p32x) {
                if (p32_nw == 8) { if (g_stat_part) DYCON_P32(conv_k3_p32x_kernel, 8, true); else DYCON_P32(conv_k3_p32x_kernel, 8, false); }
                else { if (g_stat_part) DYCON_P32(conv_k3_p32x_kernel, 4, true); else DYCON_P32(conv_k3_p32x_kernel, 4, false); }
            } else {
                if (p32_nw == 8) { if (g_stat_part) DYCON_P32(conv_k3_p32_kernel, 8, true); else DYCON_P32(conv_k3_p32_kernel, 8, false); }
                else { if (g_stat_part) DYCON_P32(conv_k3_p32_kernel, 4, true); else DYCON_P32(conv_k3_p32_kernel, 4, false); }
            }
#undef DYCON_P32
            DYCON_LAUNCH_CHECK();
            return DYCON_OK;
        }
        dim3 grid(nTiles, NT / ntb);
#define DYCON_CL(CKV, NTBV, WMV) \
    conv_k3_lds_kernel<CKV, NTBV, WMV><<<grid, 256, 0, stream>>>((const bf16*)x, (const bf16*)wfrag, bias, (bf16*)y, B, Di, Hi, Wi, Cin, Cout, NT, tz, ty, tx, accumulate)
        if (Cin == 16 || Cin == 48) {
            if (ntb == 1) DYCON_CL(16, 1, 4); else if (ntb == 2) DYCON_CL(16, 2, 4); else if (ntb == 3) DYCON_CL(16, 3, 4); else DYCON_CL(16, 4, 2);
        } else {
            static const bool w8_on = env_ll("DYCON_LDS_W8", 1) != 0;
            if (ntb == 1) DYCON_CL(32, 1, 4); else if (ntb == 2) DYCON_CL(32, 2, 4); else if (ntb == 3) DYCON_CL(32, 3, 4);
            else if (w8_on && (long long)nTiles * (NT / ntb) <= 256)     // at most one workgroup per CU: 8 waves, two per SIMD (a wave per SIMD is issue-bound)
                conv_k3_lds_kernel<32, 4, 4, 8><<<grid, 512, 0, stream>>>((const bf16*)x, (const bf16*)wfrag, bias, (bf16*)y, B, Di, Hi, Wi, Cin, Cout, NT, tz, ty, tx, accumulate);
            else DYCON_CL(32, 4, 2);
        }
#undef DYCON_CL
        DYCON_LAUNCH_CHECK();
        return DYCON_OK;
    }
    if (conv_tile_ok(dtype, mode, scatter, (long long)Di * Hi * Wi, Cin, N)) {
        const long long M = (long long)B * Di * Hi * Wi;
        const int NT = N / 16, nKC = 27 * Cin / 32;
        SplitK sk = conv_tile_plan(M, N, Cin);
        const bool split = sk.splits > 1 && workspace && ws_bytes >= (size_t)sk.splits * M * N * sizeof(float);
        dim3 grid(cdiv(M, CT_BM), N / CT_BN, split ? sk.splits : 1);
        static const bool tile64 = env_ll("DYCON_TILE_KS64", 1) != 0;
        static const bool lds_epi = env_ll("DYCON_TILE_LDS_EPI", 1) != 0;
        const int layout = split && defer_finish ? 1 : lds_epi ? 0 : 2;
        if (tile64 && Cin % 64 == 0 && sk.kc_per_split % 2 == 0)
            conv_k3_tile_kernel<2><<<grid, 256, 0, stream>>>((const bf16*)x, (const bf16*)wfrag, bias, (bf16*)y, split ? workspace : nullptr, B,
                                                             Di, Hi, Wi, Cin, N, NT, nKC, sk.kc_per_split, accumulate, layout);
        else
            conv_k3_tile_kernel<1><<<grid, 256, 0, stream>>>((const bf16*)x, (const bf16*)wfrag, bias, (bf16*)y, split ? workspace : nullptr, B,
                                                             Di, Hi, Wi, Cin, N, NT, nKC, sk.kc_per_split, accumulate, layout);
        DYCON_LAUNCH_CHECK();
        if (split && !defer_finish) {
            long long blocks = (M * N / 4 + 255) / 256;
            if (blocks > 2048) blocks = 2048;
            splitk_finish_kernel<bf16><<<(int)blocks, 256, 0, stream>>>(workspace, sk.splits, M * N, N, bias, (bf16*)y, accumulate);
            DYCON_LAUNCH_CHECK();
        }
        return DYCON_OK;
    }
    // split-K only when the caller provides the slab workspace (NULL -> single pass, same result up to fp32 summation order)
    float* ws = (workspace && ws_bytes >= dycon_conv_gemm_workspace(dtype, mode, scatter, B, Di, Hi, Wi, Cin, N)) ? workspace : nullptr;
    DYCON_DISPATCH(dtype, {
        if (scatter) launch_gemm<T, DYCON_CONV_1X1, true>(x, wfrag, bias, y, accumulate, B, Di, Hi, Wi, Cin, N, Cout, ws, defer_finish, stream);
        else if (mode == DYCON_CONV_K3) launch_gemm<T, DYCON_CONV_K3, false>(x, wfrag, bias, y, accumulate, B, Di, Hi, Wi, Cin, N, Cout, ws, defer_finish, stream);
        else if (mode == DYCON_CONV_K2S2) launch_gemm<T, DYCON_CONV_K2S2, false>(x, wfrag, bias, y, accumulate, B, Di, Hi, Wi, Cin, N, Cout, ws, defer_finish, stream);
        else launch_gemm<T, DYCON_CONV_1X1, false>(x, wfrag, bias, y, accumulate, B, Di, Hi, Wi, Cin, N, Cout, ws, defer_finish, stream);
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
    if (mode == DYCON_CONV_1X1) {   // the 2-class heads: vectorised streaming kernels
        const long long M = (long long)B * Di * Hi * Wi;
        long long blocks = (M + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (Cin == 16 && N == 2 && y_dtype == DYCON_F32 && !accumulate) {
            if (x_dtype == DYCON_F32) head_1x1_fwd_kernel<float, 16, 2><<<(int)blocks, 256, 0, stream>>>((const float*)x, w_tcn, bias, (float*)y, M);
            else head_1x1_fwd_kernel<bf16, 16, 2><<<(int)blocks, 256, 0, stream>>>((const bf16*)x, w_tcn, bias, (float*)y, M);
            DYCON_LAUNCH_CHECK();
            return DYCON_OK;
        }
        if (Cin == 2 && N == 16 && x_dtype == DYCON_F32 && !bias) {
            if (y_dtype == DYCON_F32) head_1x1_bwd_kernel<float, 2, 16><<<(int)blocks, 256, 0, stream>>>((const float*)x, w_tcn, (float*)y, M, accumulate);
            else head_1x1_bwd_kernel<bf16, 2, 16><<<(int)blocks, 256, 0, stream>>>((const float*)x, w_tcn, (bf16*)y, M, accumulate);
            DYCON_LAUNCH_CHECK();
            return DYCON_OK;
        }
    }
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
        long long rps = 1024;
        p.splits = (int)((M + rps - 1) / rps);
        p.rows_per_split = rps;
    }
    return p;
}

extern "C" size_t dycon_colsum_workspace(long long rows, int C);
extern "C" int dycon_colsum(const void* x, int dtype, float* out, long long rows, int C, float* workspace, size_t ws_bytes,
                            dycon_stream_t stream);

// k=3 bf16 plan: tiles and voxel-tile splits
struct WgradK3Plan { int tilesZ, tilesY, tilesX, nTiles, gx, nCoBlk, NT, splits; };
static WgradK3Plan wgrad_k3_plan(int B, int D, int H, int W, int Cin, int Cout) {
    WgradK3Plan p;
    p.tilesZ = cdiv(D, WG_TZ); p.tilesY = cdiv(H, WG_TY); p.tilesX = cdiv(W, WG_TX);
    p.nTiles = B * p.tilesZ * p.tilesY * p.tilesX;
    p.NT = Cout >= 64 ? 4 : Cout / 16;
    p.nCoBlk = cdiv(Cout, 16 * p.NT);
    p.gx = ((Cin + 15) / 16) * p.nCoBlk;
    const long long L = 27LL * Cin * Cout;
    long long s = 2048 / p.gx;
    static const long long slab_mb = env_ll("DYCON_WGRAD_SLAB_MB", 24);
    const long long cap = (slab_mb << 20) / (4 * L) > 0 ? (slab_mb << 20) / (4 * L) : 1;   // partial slabs capped at ~24 MB (fewer splits = fewer workgroups = slower: measured)
    if (s > cap) s = cap;
    if (s > p.nTiles) s = p.nTiles;
    if (s < 1) s = 1;
    p.splits = (int)s;
    return p;
}
static bool wgrad_k2_ok(int mode, int Cin, int Cout) {
    return mode == DYCON_CONV_K2S2 && Cin % 16 == 0 && Cout % 16 == 0 && (Cout == 16 || Cout == 32 || Cout % 64 == 0);
}
static WgradK3Plan wgrad_k2_plan(int B, int Di, int Hi, int Wi, int Cin, int Cout) {   // tiles over the LO (half) grid
    WgradK3Plan p;
    p.tilesZ = cdiv(Di / 2, WG_TZ); p.tilesY = cdiv(Hi / 2, WG_TY); p.tilesX = cdiv(Wi / 2, WG_TX);
    p.nTiles = B * p.tilesZ * p.tilesY * p.tilesX;
    p.NT = Cout >= 64 ? 4 : Cout / 16;
    p.nCoBlk = cdiv(Cout, 16 * p.NT);
    p.gx = (Cin / 16) * p.nCoBlk;
    const long long L = 8LL * Cin * Cout;
    long long s = 2048 / p.gx;
    const long long cap = (16LL << 20) / (4 * L);
    if (s > cap) s = cap;
    if (s > p.nTiles) s = p.nTiles;
    if (s < 1) s = 1;
    p.splits = (int)s;
    return p;
}
static bool wgrad_1x1_ok(int mode, int Cin, int Cout) {
    return mode == DYCON_CONV_1X1 && Cin % W1_CI == 0 && Cout % 16 == 0 && (Cout == 16 || Cout == 32 || Cout % 64 == 0);
}
static WgradK3Plan wgrad_1x1_plan(long long M, int Cin, int Cout) {
    WgradK3Plan p{};
    p.nTiles = (int)((M + W1_ROWS - 1) / W1_ROWS);
    p.NT = Cout >= 64 ? 4 : Cout / 16;
    p.nCoBlk = cdiv(Cout, 16 * p.NT);
    p.gx = (Cin / W1_CI) * p.nCoBlk;
    long long s_ = 512 / p.gx;
    if (s_ > p.nTiles) s_ = p.nTiles;
    if (s_ < 1) s_ = 1;
    p.splits = (int)s_;
    return p;
}
static bool wgrad_k3_ok(int mode, int Cin, int Cout) {
    return mode == DYCON_CONV_K3 && (Cin % 16 == 0 || Cin == 1) && Cout % 16 == 0 && (Cout == 16 || Cout == 32 || Cout % 64 == 0);
}

static int wgrad_c1_wgs(int B, int Di, int Hi, int Wi) {      // persistent workgroups (= partial rows) of wgrad_k3_c1_kernel
    const int nTiles = B * cdiv(Di, CL_TZ) * cdiv(Hi, CL_TY) * cdiv(Wi, CL_TX);
    return 8 * min(cdiv(nTiles, 8), 64);
}

// First layer's weight + bias gradient with the data gradient of the normalisation that follows the first convolution formed on
// load (wgrad_k3_c1_kernel<true>): gy = gradient w.r.t. that normalisation's output, z = its input, stats / ab = its forward
// statistics and the {A, B} sums dycon_norm_bwd_stats left behind.  bf16, one input channel, 16 output channels.
extern "C" size_t dycon_conv1_wgrad_normbwd_workspace(int B, int D, int H, int W) {
    return (size_t)wgrad_c1_wgs(B, D, H, W) * (27 * 16 + 16) * sizeof(float);
}
extern "C" int dycon_conv1_wgrad_normbwd(const void* x, const void* z, const void* gy, int B, int D, int H, int W, int Nb, int G,
                                         const float* stats, const float* gamma, const float* beta, int relu,
                                         const float* chan_scale, const float* ab, float* dw, float* dbias, long long s_t,
                                         long long s_c, long long s_n, float* workspace, size_t ws_bytes, dycon_stream_t stream) {
    DYCON_REQUIRE(x && z && gy && stats && ab && dw && workspace, "conv1_wgrad_normbwd: null pointer");
    DYCON_REQUIRE(B > 0 && B <= 16 && D > 0 && H > 0 && W > 0, "conv1_wgrad_normbwd: bad shape (B <= 16)");
    DYCON_REQUIRE((Nb == B || Nb == 1) && G > 0 && 16 % G == 0, "conv1_wgrad_normbwd: Nb must be B or 1, G must divide 16");
    DYCON_REQUIRE(ws_bytes >= dycon_conv1_wgrad_normbwd_workspace(B, D, H, W), "conv1_wgrad_normbwd: workspace too small");
    const int tz = cdiv(D, CL_TZ), ty = cdiv(H, CL_TY), tx = cdiv(W, CL_TX);
    const int nTiles = B * tz * ty * tx, wgs = wgrad_c1_wgs(B, D, H, W), L = 27 * 16;
    float* bp = dbias ? workspace + (size_t)wgs * L : nullptr;
    C1NormBwd nb;
    nb.Z = (const bf16*)z; nb.stats = stats; nb.gamma = gamma; nb.beta = beta; nb.ab = ab; nb.chan_scale = chan_scale;
    nb.Nb = Nb; nb.G = G; nb.relu = relu;
    nb.inv_cnt = 1.f / ((float)((long long)D * H * W * (Nb == 1 ? B : 1)) * (float)(16 / G));
    wgrad_k3_c1_kernel<true><<<wgs, 256, 0, stream>>>((const bf16*)x, (const bf16*)gy, workspace, bp, B, D, H, W, tz, ty, tx, nTiles, nb);
    DYCON_LAUNCH_CHECK();
    launch_reduce_partials(workspace, wgs, L, dw, 1, 16, s_t, s_c, s_n, stream, bp, dbias);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}


extern "C" size_t dycon_conv_wgrad_workspace(int mode, int B, int Di, int Hi, int Wi, int Cin, int Cout) {
    const WgradPlan p = wgrad_plan(mode, B, Di, Hi, Wi, Cin, Cout);
    size_t need = (size_t)p.splits * p.L * sizeof(float);
    if (mode == DYCON_CONV_K3 && Cin == 1 && Cout == 16) {
        const size_t n3 = (size_t)wgrad_c1_wgs(B, Di, Hi, Wi) * (p.L + Cout) * sizeof(float);
        if (n3 > need) need = n3;
    }
    if (wgrad_k3_ok(mode, Cin, Cout) || wgrad_k2_ok(mode, Cin, Cout)) {
        const WgradK3Plan k = mode == DYCON_CONV_K3 ? wgrad_k3_plan(B, Di, Hi, Wi, Cin, Cout) : wgrad_k2_plan(B, Di, Hi, Wi, Cin, Cout);
        const size_t n2 = ((size_t)k.splits * p.L + (size_t)k.splits * Cout) * sizeof(float);
        if (n2 > need) need = n2;
    }
    int Do, Ho, Wo;
    row_grid(mode, Di, Hi, Wi, Do, Ho, Wo);
    if (wgrad_1x1_ok(mode, Cin, Cout)) {
        const WgradK3Plan k = wgrad_1x1_plan((long long)B * Di * Hi * Wi, Cin, Cout);
        const size_t n2 = ((size_t)k.splits * p.L + (size_t)k.splits * Cout) * sizeof(float);
        if (n2 > need) need = n2;
    }
    return need + dycon_colsum_workspace((long long)B * Do * Ho * Wo, Cout) + 64;   // room for the un-fused bias gradient
}

// The whole backward of block_one in one pass (first_block_bwd_kernel): dW / db of the first convolution, dgamma / dbeta of the
// normalisation that follows it.  bf16, one input channel, 16 output channels, B <= 16.
extern "C" size_t dycon_first_block_bwd_workspace(int B, int D, int H, int W) {
    return (size_t)B * wgrad_c1_wgs(B, D, H, W) * FB_ROW * sizeof(float) + (size_t)B * FB_ROW * sizeof(double) + 64;
}
extern "C" int dycon_first_block_bwd(const void* x, const void* z, const void* gy, int B, int D, int H, int W, int Nb, int G,
                                     const float* stats, const float* gamma, const float* beta, int relu,
                                     const float* chan_scale, float* dgamma, float* dbeta, float* dw, float* dbias, long long s_t,
                                     long long s_c, long long s_n, float* workspace, size_t ws_bytes, dycon_stream_t stream) {
    DYCON_REQUIRE(x && z && gy && stats && dw && workspace, "first_block_bwd: null pointer");
    DYCON_REQUIRE(B > 0 && B <= 16 && D > 0 && H > 0 && W > 0, "first_block_bwd: bad shape (B <= 16)");
    DYCON_REQUIRE((Nb == B || Nb == 1) && G > 0 && 16 % G == 0, "first_block_bwd: Nb must be B or 1, G must divide 16");
    DYCON_REQUIRE(ws_bytes >= dycon_first_block_bwd_workspace(B, D, H, W), "first_block_bwd: workspace too small");
    const int tz = cdiv(D, CL_TZ), ty = cdiv(H, CL_TY), tx = cdiv(W, CL_TX);
    const int nTiles = B * tz * ty * tx, wgs = wgrad_c1_wgs(B, D, H, W);
    double* tot = reinterpret_cast<double*>(workspace + (((size_t)B * wgs * FB_ROW + 1) & ~(size_t)1));
    FbFwd f;
    f.stats = stats; f.gamma = gamma; f.beta = beta; f.chan_scale = chan_scale; f.Nb = Nb; f.G = G; f.relu = relu;
    first_block_bwd_kernel<<<wgs, 256, 0, stream>>>((const bf16*)x, (const bf16*)z, (const bf16*)gy, workspace, B, D, H, W, tz, ty, tx, nTiles, f, tot);
    DYCON_LAUNCH_CHECK();
    first_block_reduce_kernel<<<dim3(cdiv(FB_ROW, 256), B, 32), 256, 0, stream>>>(workspace, wgs, tot);
    DYCON_LAUNCH_CHECK();
    first_block_finalize_kernel<<<1, 256, 0, stream>>>(tot, B, (long long)D * H * W, f, dgamma, dbeta, dw, dbias, s_t, s_c, s_n);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
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

extern "C" int dycon_conv_wgrad(const void* x, int x_dtype, const void* gy, int g_dtype, float* dw, float* dbias, int mode, int B,
                                int Di, int Hi, int Wi, int Cin, int Cout, long long s_t, long long s_c, long long s_n,
                                float* workspace, size_t ws_bytes, dycon_stream_t stream) {
    DYCON_REQUIRE(x && gy && dw && workspace, "conv_wgrad: null pointer");
    DYCON_REQUIRE(B > 0 && Di > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "conv_wgrad: bad shape");
    DYCON_REQUIRE(mode >= 0 && mode <= 2, "conv_wgrad: bad mode %d", mode);
    DYCON_REQUIRE(ws_bytes >= dycon_conv_wgrad_workspace(mode, B, Di, Hi, Wi, Cin, Cout), "conv_wgrad: workspace too small");
    const WgradPlan p = wgrad_plan(mode, B, Di, Hi, Wi, Cin, Cout);
    if (x_dtype == DYCON_BF16 && g_dtype == DYCON_BF16 && wgrad_k3_ok(mode, Cin, Cout)) {
        const WgradK3Plan k = wgrad_k3_plan(B, Di, Hi, Wi, Cin, Cout);
        static const bool wc1_on = env_ll("DYCON_WGRAD_C1", 1) != 0;
        if (wc1_on && Cin == 1 && Cout == 16) {      // first layer: transposed product, persistent workgroups (wgrad_k3_c1_kernel)
            const int tz = cdiv(Di, CL_TZ), ty = cdiv(Hi, CL_TY), tx = cdiv(Wi, CL_TX);
            const int nTiles = B * tz * ty * tx;
            const int wgs = wgrad_c1_wgs(B, Di, Hi, Wi);
            float* bp = dbias ? workspace + (size_t)wgs * p.L : nullptr;
            wgrad_k3_c1_kernel<false><<<wgs, 256, 0, stream>>>((const bf16*)x, (const bf16*)gy, workspace, bp, B, Di, Hi, Wi, tz, ty, tx, nTiles, C1NormBwd{});
            DYCON_LAUNCH_CHECK();
            launch_reduce_partials(workspace, wgs, p.L, dw, Cin, Cout, s_t, s_c, s_n, stream, bp, dbias);
            DYCON_LAUNCH_CHECK();
            return DYCON_OK;
        }
        float* bpart = dbias ? workspace + (size_t)k.splits * p.L : nullptr;
        dim3 grid(k.gx, k.splits);
#define DYCON_WK3(NTV, C1) \
    wgrad_k3_bf16_kernel<NTV, C1><<<grid, 256, 0, stream>>>((const bf16*)x, (const bf16*)gy, workspace, bpart, B, Di, Hi, Wi, Cin, Cout, k.nCoBlk, k.nTiles, k.tilesZ, k.tilesY, k.tilesX)
        static const bool wg_w8 = env_ll("DYCON_WGRAD_W8", 1) != 0;
        if (wg_w8 && Cin != 1 && k.NT == 4 && (long long)k.gx * k.splits <= 256)      // at most one workgroup per CU: 8 waves, two per SIMD
            wgrad_k3_bf16_kernel<4, false, 8><<<grid, 512, 0, stream>>>((const bf16*)x, (const bf16*)gy, workspace, bpart, B, Di, Hi, Wi, Cin, Cout, k.nCoBlk, k.nTiles, k.tilesZ, k.tilesY, k.tilesX);
        else if (Cin == 1) { if (k.NT == 1) DYCON_WK3(1, true); else if (k.NT == 2) DYCON_WK3(2, true); else DYCON_WK3(4, true); }
        else if (k.NT == 1) DYCON_WK3(1, false);
        else if (k.NT == 2) DYCON_WK3(2, false);
        else DYCON_WK3(4, false);
#undef DYCON_WK3
        DYCON_LAUNCH_CHECK();
        launch_reduce_partials(workspace, k.splits, p.L, dw, Cin, Cout, s_t, s_c, s_n, stream, bpart, dbias);   // weights + bias: one launch
        DYCON_LAUNCH_CHECK();
        return DYCON_OK;
    }
    if (x_dtype == DYCON_BF16 && g_dtype == DYCON_BF16 && wgrad_k2_ok(mode, Cin, Cout)) {
        const WgradK3Plan k = wgrad_k2_plan(B, Di, Hi, Wi, Cin, Cout);
        float* bpart = dbias ? workspace + (size_t)k.splits * p.L : nullptr;
        dim3 grid(k.gx, k.splits);
#define DYCON_WK2(NTV) \
    wgrad_k2s2_bf16_kernel<NTV><<<grid, 256, 0, stream>>>((const bf16*)x, (const bf16*)gy, workspace, bpart, B, Di / 2, Hi / 2, Wi / 2, Cin, Cout, k.nCoBlk, k.nTiles, k.tilesZ, k.tilesY, k.tilesX)
        if (k.NT == 1) DYCON_WK2(1);
        else if (k.NT == 2) DYCON_WK2(2);
        else DYCON_WK2(4);
#undef DYCON_WK2
        DYCON_LAUNCH_CHECK();
        launch_reduce_partials(workspace, k.splits, p.L, dw, Cin, Cout, s_t, s_c, s_n, stream, bpart, dbias);   // weights + bias: one launch
        DYCON_LAUNCH_CHECK();
        return DYCON_OK;
    }
    if (x_dtype == DYCON_BF16 && g_dtype == DYCON_BF16 && wgrad_1x1_ok(mode, Cin, Cout)) {
        const long long M = (long long)B * Di * Hi * Wi;
        const WgradK3Plan k = wgrad_1x1_plan(M, Cin, Cout);
        float* bpart = dbias ? workspace + (size_t)k.splits * p.L : nullptr;
        dim3 grid(k.gx, k.splits);
#define DYCON_W11(NTV) \
    wgrad_1x1_bf16_kernel<NTV><<<grid, 256, 0, stream>>>((const bf16*)x, (const bf16*)gy, workspace, bpart, M, Cin, Cout, k.nCoBlk, k.nTiles)
        if (k.NT == 1) DYCON_W11(1);
        else if (k.NT == 2) DYCON_W11(2);
        else DYCON_W11(4);
#undef DYCON_W11
        DYCON_LAUNCH_CHECK();
        launch_reduce_partials(workspace, k.splits, p.L, dw, Cin, Cout, s_t, s_c, s_n, stream, bpart, dbias);
        DYCON_LAUNCH_CHECK();
        return DYCON_OK;
    }
    if (mode == DYCON_CONV_1X1 && Cin == 16 && Cout == 2 && g_dtype == DYCON_F32) {
        const long long M = (long long)B * Di * Hi * Wi;
        int blocks = p.splits;                      // partial rows available in the workspace (>= M/1024)
        if (blocks > 1024) blocks = 1024;
        if (x_dtype == DYCON_F32) wgrad_1x1_skinny_kernel<float, float, 16, 2><<<blocks, 256, 0, stream>>>((const float*)x, (const float*)gy, workspace, M);
        else wgrad_1x1_skinny_kernel<bf16, float, 16, 2><<<blocks, 256, 0, stream>>>((const bf16*)x, (const float*)gy, workspace, M);
        DYCON_LAUNCH_CHECK();
        launch_reduce_partials(workspace, blocks, p.L, dw, Cin, Cout, s_t, s_c, s_n, stream);
        DYCON_LAUNCH_CHECK();
        if (dbias) {
            float* cws = workspace + (size_t)p.splits * p.L;
            return dycon_colsum(gy, g_dtype, dbias, M, Cout, cws, dycon_colsum_workspace(M, Cout), stream);
        }
        return DYCON_OK;
    }
    if (x_dtype == DYCON_F32 && g_dtype == DYCON_F32) launch_wgrad<float, float>(x, gy, workspace, mode, p, B, Di, Hi, Wi, Cin, Cout, stream);
    else if (x_dtype == DYCON_BF16 && g_dtype == DYCON_BF16) launch_wgrad<bf16, bf16>(x, gy, workspace, mode, p, B, Di, Hi, Wi, Cin, Cout, stream);
    else if (x_dtype == DYCON_BF16 && g_dtype == DYCON_F32) launch_wgrad<bf16, float>(x, gy, workspace, mode, p, B, Di, Hi, Wi, Cin, Cout, stream);
    else if (x_dtype == DYCON_F32 && g_dtype == DYCON_BF16) launch_wgrad<float, bf16>(x, gy, workspace, mode, p, B, Di, Hi, Wi, Cin, Cout, stream);
    else { dycon_set_error("conv_wgrad: bad dtypes"); return DYCON_ERR_INVALID; }
    DYCON_LAUNCH_CHECK();
    launch_reduce_partials(workspace, p.splits, p.L, dw, Cin, Cout, s_t, s_c, s_n, stream);
    DYCON_LAUNCH_CHECK();
    if (dbias) {   // un-fused bias gradient: column sums of gy, partials behind the weight partials
        int Do, Ho, Wo;
        row_grid(mode, Di, Hi, Wi, Do, Ho, Wo);
        const long long rows = (long long)B * Do * Ho * Wo;
        float* cws = workspace + (size_t)p.splits * p.L;
        return dycon_colsum(gy, g_dtype, dbias, rows, Cout, cws, dycon_colsum_workspace(rows, Cout), stream);
    }
    return DYCON_OK;
}

static void colsum_plan(long long rows, int C, int& blocks, long long& rpb) {
    long long b = (rows + 255) / 256;
    if (b > 2048) b = 2048;
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
    launch_reduce_partials(workspace, blocks, C, out, 1, C, 0, 0, 1, stream);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}
