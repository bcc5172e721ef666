// Data-movement and pointwise kernels of the DyCON step (all HBM-bound streams, NDHWC).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (no state tensor: the same (seed, offset) regenerates a mask in backward)
// ------------------------------------------------------------------------------------------------
struct Philox {
    uint32_t k0, k1;
    __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __device__ uint4 operator()(uint64_t ctr, uint32_t stream_hi = 0) const {
        uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = stream_hi, c3 = 0x9E3779B9u;
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b, n3 = (uint32_t)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

// ------------------------------------------------------------------------------------------------
// MaxPool3d(2): first maximum in (dz,dy,dx) scan order wins (ATen CPU/CUDA tie rule; matters because
// whole windows are 0 after ReLU)
// ------------------------------------------------------------------------------------------------
// one thread = one output voxel x VN channels (16-byte loads/stores)
template <typename T>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ X, T* __restrict__ Y, uint8_t* __restrict__ idx, int B, int D, int H,
                                    int W, int C, long long total) {
    constexpr int VN = Vec16<T>::N;
    const int Do = D / 2, Ho = H / 2, Wo = W / 2, CG = C / VN;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        long long q = i / CG;
        const int x = (int)(q % Wo); q /= Wo;
        const int y = (int)(q % Ho); q /= Ho;
        const int z = (int)(q % Do);
        const int b = (int)(q / Do);
        float best[VN];
        int bi[VN];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const long long v = (((long long)b * D + 2 * z + (t >> 2)) * H + 2 * y + ((t >> 1) & 1)) * W + 2 * x + (t & 1);
            const Vec16<T> f = ld16(X + v * C + cg * VN);
#pragma unroll
            for (int k = 0; k < VN; ++k) {
                const float fv = f.get(k);
                if (t == 0 || fv > best[k] || (fv != fv && best[k] == best[k])) { best[k] = fv; bi[k] = t; }
            }
        }
        const long long o = ((((long long)b * Do + z) * Ho + y) * Wo + x) * C + cg * VN;
        Vec16<T> out;
#pragma unroll
        for (int k = 0; k < VN; ++k) { out.set(k, best[k]); idx[o + k] = (uint8_t)bi[k]; }
        st16(Y + o, out);
    }
}

template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ GY, const uint8_t* __restrict__ idx, T* __restrict__ GX, int B, int D,
                                    int H, int W, int C, long long total) {
    constexpr int VN = Vec16<T>::N;
    const int Do = D / 2, Ho = H / 2, Wo = W / 2, CG = C / VN;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        long long q = i / CG;
        const int x = (int)(q % W); q /= W;
        const int y = (int)(q % H); q /= H;
        const int z = (int)(q % D);
        const int b = (int)(q / D);
        Vec16<T> g;
        g.v = decltype(g.v){};
        if ((z >> 1) < Do && (y >> 1) < Ho && (x >> 1) < Wo) {
            const long long o = ((((long long)b * Do + (z >> 1)) * Ho + (y >> 1)) * Wo + (x >> 1)) * C + cg * VN;
            const int t = ((z & 1) << 2) | ((y & 1) << 1) | (x & 1);
            const Vec16<T> gy = ld16(GY + o);
#pragma unroll
            for (int k = 0; k < VN; ++k) g.set(k, idx[o + k] == t ? gy.get(k) : 0.f);
        }
        st16(GX + ((((long long)b * D + z) * H + y) * W + x) * C + cg * VN, g);
    }
}

// ------------------------------------------------------------------------------------------------
// trilinear resize, ATen's source-index rule (UpSample.h area_pixel_compute_source_index)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tri_src(int o, int in, int out, int align, int& i0, int& i1, float& lam) {
    float src;
    if (align) {
        const float sc = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
        src = sc * (float)o;
    } else {
        const float sc = (float)in / (float)out;
        src = sc * ((float)o + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
    }
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    lam = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
}

// element index -> (channel group, x, y, z, b).  32-bit arithmetic whenever the index fits (every shape of the step): a 64-bit
// division by a runtime divisor is ~60 vector instructions, five of them per element were most of what these kernels executed
// (tools/isa_loop_mix.py: 115 division-sequence instructions of 670 in the loop of trilinear_fwd_kernel).
__device__ __forceinline__ void decode5(long long i, int CG, int W, int H, int D, int& cg, int& x, int& y, int& z, int& b) {
    if (i < (1ll << 31)) {
        unsigned q = (unsigned)i;
        unsigned t = q / (unsigned)CG; cg = (int)(q - t * CG); q = t;
        t = q / (unsigned)W; x = (int)(q - t * W); q = t;
        t = q / (unsigned)H; y = (int)(q - t * H); q = t;
        t = q / (unsigned)D; z = (int)(q - t * D);
        b = (int)t;
    } else {
        cg = (int)(i % CG);
        long long q = i / CG;
        x = (int)(q % W); q /= W;
        y = (int)(q % H); q /= H;
        z = (int)(q % D);
        b = (int)(q / D);
    }
}

// one thread = one output voxel x VN channels: coordinates once, 8 x 16-byte taps
template <typename T>
__global__ void trilinear_fwd_kernel(const T* __restrict__ X, T* __restrict__ Y, int B, int Di, int Hi, int Wi, int Do, int Ho,
                                     int Wo, int C, int ldy, int coff, int align, long long total) {
    constexpr int VN = Vec16<T>::N;
    const int CG = C / VN;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cg, x, y, z, b;
        decode5(i, CG, Wo, Ho, Do, cg, x, y, z, b);
        int z0, z1, y0, y1, x0, x1;
        float lz, ly, lx;
        tri_src(z, Di, Do, align, z0, z1, lz);
        tri_src(y, Hi, Ho, align, y0, y1, ly);
        tri_src(x, Wi, Wo, align, x0, x1, lx);
        const T* xb = X + (long long)b * Di * Hi * Wi * C + cg * VN;
        auto at = [&](int zz, int yy, int xx) { return ld16(xb + (long long)((zz * Hi + yy) * Wi + xx) * C); };
        const Vec16<T> a000 = at(z0, y0, x0), a001 = at(z0, y0, x1), a010 = at(z0, y1, x0), a011 = at(z0, y1, x1);
        const Vec16<T> a100 = at(z1, y0, x0), a101 = at(z1, y0, x1), a110 = at(z1, y1, x0), a111 = at(z1, y1, x1);
        const float w0z = 1.f - lz, w0y = 1.f - ly, w0x = 1.f - lx;
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < VN; ++k) {   // same association order as ATen's upsample_trilinear3d
            const float v = w0z * (w0y * (w0x * a000.get(k) + lx * a001.get(k)) + ly * (w0x * a010.get(k) + lx * a011.get(k))) +
                            lz * (w0y * (w0x * a100.get(k) + lx * a101.get(k)) + ly * (w0x * a110.get(k) + lx * a111.get(k)));
            o.set(k, v);
        }
        st16(Y + ((((long long)b * Do + z) * Ho + y) * Wo + x) * ldy + coff + cg * VN, o);
    }
}

__device__ __forceinline__ float tri_w(int o, int i, int in, int out, int align) {
    int i0, i1;
    float lam;
    tri_src(o, in, out, align, i0, i1, lam);
    return (i == i0 ? 1.f - lam : 0.f) + (i == i1 ? lam : 0.f);
}
__device__ __forceinline__ void tri_range(int i, int in, int out, int align, int& lo, int& hi) {
    // superset of the outputs whose two taps can include input i (weights outside are exactly 0): an output o reads inputs
    // floor(src(o)) and floor(src(o)) + 1, so it can touch i only if src(o) lies in (i - 1, i + 1).  align: src = o (in-1)/(out-1);
    // otherwise src = (o + 0.5) in/out - 0.5 (clamped at 0, which only adds outputs to i = 0).  One extra output on either side
    // absorbs the rounding of the bounds: 6 candidates per axis at scale 2 (4 contribute), where the old +-(r/2 + 1) margin gave 9.
    if (align && in > 1) {
        const float r = (float)(out - 1) / (float)(in - 1);
        lo = (int)floorf(((float)i - 1.f) * r) - 1;
        hi = (int)ceilf(((float)i + 1.f) * r) + 1;
    } else {
        const float r = (float)out / (float)in;
        lo = (int)floorf(((float)i - 0.5f) * r - 0.5f) - 1;
        hi = (int)ceilf(((float)i + 1.5f) * r - 0.5f) + 1;
    }
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
}

// gather form of the adjoint: deterministic, no atomics.  One thread = one input voxel x VN channels; the per-axis weights of the
// (few) contributing outputs are evaluated once per axis.
template <typename T>
__global__ void trilinear_bwd_kernel(const T* __restrict__ GY, T* __restrict__ GX, int B, int Di, int Hi, int Wi, int Do, int Ho,
                                     int Wo, int C, int ldy, int coff, int align, long long total) {
    constexpr int VN = Vec16<T>::N;
    constexpr int MAXO = 12;                       // candidate outputs per axis held in registers (tri_range: 6 at scale 2, 12 at 4.5)
    const int CG = C / VN;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cg, x, y, z, b;
        decode5(i, CG, Wi, Hi, Di, cg, x, y, z, b);
        int zl, zh, yl, yh, xl, xh;
        tri_range(z, Di, Do, align, zl, zh);
        tri_range(y, Hi, Ho, align, yl, yh);
        tri_range(x, Wi, Wo, align, xl, xh);
        float acc[VN];
#pragma unroll
        for (int k = 0; k < VN; ++k) acc[k] = 0.f;
        // x-axis weights of the candidate outputs once per thread (statically indexed registers): evaluated inside the loop nest they
        // were recomputed for every (oz, oy) pair -- 144 of the 189 coordinate computations per thread at scale 2
        float wxs[MAXO];
        const bool xfit = xh - xl < MAXO;
#pragma unroll
        for (int k = 0; k < MAXO; ++k) wxs[k] = (xfit && xl + k <= xh) ? tri_w(xl + k, x, Wi, Wo, align) : 0.f;
        float wys[MAXO];                               // ... and the y-axis weights (they were re-evaluated for every oz)
        const bool yfit = yh - yl < MAXO;
#pragma unroll
        for (int k = 0; k < MAXO; ++k) wys[k] = (yfit && yl + k <= yh) ? tri_w(yl + k, y, Hi, Ho, align) : 0.f;
        if (xfit && yfit) {                            // every scale the networks use: both inner axes walk register-held weights
            for (int oz = zl; oz <= zh; ++oz) {
                const float wz = tri_w(oz, z, Di, Do, align);
                if (wz == 0.f) continue;
                const T* plane = GY + ((((long long)b * Do + oz) * Ho + yl) * Wo + xl) * ldy + coff + cg * VN;
#pragma unroll
                for (int ky = 0; ky < MAXO; ++ky) {
                    const float wzy = wz * wys[ky];
                    if (wzy == 0.f) continue;
                    const T* row = plane + (long long)ky * Wo * ldy;
#pragma unroll
                    for (int k = 0; k < MAXO; ++k) {
                        const float w = wzy * wxs[k];
                        if (w == 0.f) continue;
                        const Vec16<T> g = ld16(row + (long long)k * ldy);
#pragma unroll
                        for (int e = 0; e < VN; ++e) acc[e] += w * g.get(e);
                    }
                }
            }
        } else {                                       // very large scale factors: evaluate in place
            for (int oz = zl; oz <= zh; ++oz) {
                const float wz = tri_w(oz, z, Di, Do, align);
                if (wz == 0.f) continue;
                for (int oy = yl; oy <= yh; ++oy) {
                    const float wzy = wz * tri_w(oy, y, Hi, Ho, align);
                    if (wzy == 0.f) continue;
                    const T* row = GY + ((((long long)b * Do + oz) * Ho + oy) * Wo + xl) * ldy + coff + cg * VN;
                    for (int ox = xl; ox <= xh; ++ox) {
                        const float w = wzy * tri_w(ox, x, Wi, Wo, align);
                        if (w == 0.f) continue;
                        const Vec16<T> g = ld16(row + (long long)(ox - xl) * ldy);
#pragma unroll
                        for (int e = 0; e < VN; ++e) acc[e] += w * g.get(e);
                    }
                }
            }
        }
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < VN; ++k) o.set(k, acc[k]);
        st16(GX + i * VN, o);
    }
}

// ------------------------------------------------------------------------------------------------
// pointwise
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ S, int lds, int soff, T* __restrict__ Dst, int ldd, int doff,
                                     long long rows, int C) {
    constexpr int VN = Vec16<T>::N;
    if (C % VN == 0 && lds % VN == 0 && ldd % VN == 0 && soff % VN == 0 && doff % VN == 0) {
        const int CG = C / VN;
        const long long total = rows * CG;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            const long long r = i / CG;
            const int c = (int)(i % CG) * VN;
            st16(Dst + r * ldd + doff + c, ld16(S + r * lds + soff + c));
        }
        return;
    }
    const long long total = rows * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i % C);
        Dst[r * ldd + doff + c] = S[r * lds + soff + c];
    }
}

template <typename T>
__global__ void scale_channels_kernel(const T* __restrict__ X, const float* __restrict__ scale, T* __restrict__ Y, long long V,
                                      int C, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long b = i / (V * C);
        stf(Y + i, ldf(X + i) * scale[b * C + c]);
    }
}

// normalization='none' blocks of the V-Net (VNet.py:23-24, 87, 114): y = relu(z) * chan_scale[b, c] + skip
template <typename T>
__global__ void relu_fwd_kernel(const T* __restrict__ Z, const T* __restrict__ skip, const float* __restrict__ cs, T* __restrict__ Y,
                                long long V, int C, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float v = fmaxf(ldf(Z + i), 0.f);
        if (cs) v *= cs[(i / (V * C)) * C + (int)(i % C)];
        if (skip) v += ldf(skip + i);
        stf(Y + i, v);
    }
}
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ Z, const T* __restrict__ GY, const float* __restrict__ cs, T* __restrict__ GZ,
                                long long V, int C, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float g = ldf(Z + i) > 0.f ? ldf(GY + i) : 0.f;
        if (cs) g *= cs[(i / (V * C)) * C + (int)(i % C)];
        stf(GZ + i, g);
    }
}

template <typename T>
__global__ void mul_mask_kernel(const T* __restrict__ X, const float* __restrict__ mask, float inv_keep, T* __restrict__ Y,
                                long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        stf(Y + i, ldf(X + i) * mask[i] * inv_keep);
}

template <typename T>
__global__ void dropout_philox_kernel(const T* __restrict__ X, T* __restrict__ Y, long long n, float p, uint64_t seed,
                                      uint64_t offset) {
    const Philox ph(seed);
    const float inv = 1.f / (1.f - p);
    const long long n4 = (n + 3) / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const uint4 r = ph(offset + (uint64_t)i, 1);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long e = i * 4 + k;
            if (e < n) stf(Y + e, u01(rr[k]) > p ? ldf(X + e) * inv : 0.f);
        }
    }
}

__global__ void channel_mask_philox_kernel(float* __restrict__ scale, long long n, float p, uint64_t seed, uint64_t offset) {
    const Philox ph(seed);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 r = ph(offset + (uint64_t)i, 2);
    scale[i] = u01(r.x) > p ? 1.f / (1.f - p) : 0.f;
}

template <typename T>
__global__ void add_noise_kernel(const T* __restrict__ X, const float* __restrict__ noise, T* __restrict__ Y, long long n,
                                 float sigma, float clip, uint64_t seed, uint64_t offset) {
    const Philox ph(seed);
    const long long n4 = (n + 3) / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float z[4];
        if (noise == nullptr) {
            const uint4 r = ph(offset + (uint64_t)i, 3);
            // Box-Muller, two pairs
            const float r0 = sqrtf(-2.f * __logf(u01(r.x))), r1 = sqrtf(-2.f * __logf(u01(r.z)));
            float s0, c0, s1, c1;
            __sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
            __sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
            z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long e = i * 4 + k;
            if (e >= n) continue;
            float d;
            if (noise) d = noise[e];
            else d = fminf(fmaxf(z[k] * sigma, -clip), clip);
            stf(Y + e, ldf(X + e) + d);
        }
    }
}

template <typename T>
__global__ void tanh_kernel(const T* __restrict__ X, float* __restrict__ Y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        Y[i] = tanhf(ldf(X + i));
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ X, TO* __restrict__ Y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        stf(Y + i, ldf(X + i));
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ A, const T* __restrict__ Bv, T* __restrict__ Y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        stf(Y + i, ldf(A + i) + (Bv ? ldf(Bv + i) : 0.f));
}

// ------------------------------------------------------------------------------------------------
static inline int sgrid(long long n) {
    long long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int dycon_maxpool2_fwd(const void* x, void* y, uint8_t* idx, int dtype, int B, int D, int H, int W, int C,
                                  dycon_stream_t stream) {
    DYCON_REQUIRE(x && y && idx && B > 0 && D > 1 && H > 1 && W > 1 && C > 0, "maxpool2_fwd: bad arguments");
    DYCON_REQUIRE(C % (dtype == DYCON_BF16 ? 8 : 4) == 0, "maxpool2_fwd: C=%d must be a multiple of the 16-byte vector width", C);
    const long long total = (long long)B * (D / 2) * (H / 2) * (W / 2) * (C / (dtype == DYCON_BF16 ? 8 : 4));
    DYCON_DISPATCH(dtype, { maxpool2_fwd_kernel<T><<<sgrid(total), 256, 0, stream>>>((const T*)x, (T*)y, idx, B, D, H, W, C, total); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_maxpool2_bwd(const void* gy, const uint8_t* idx, void* gx, int dtype, int B, int D, int H, int W, int C,
                                  dycon_stream_t stream) {
    DYCON_REQUIRE(gy && gx && idx && B > 0 && D > 1 && H > 1 && W > 1 && C > 0, "maxpool2_bwd: bad arguments");
    DYCON_REQUIRE(C % (dtype == DYCON_BF16 ? 8 : 4) == 0, "maxpool2_bwd: C=%d must be a multiple of the 16-byte vector width", C);
    const long long total = (long long)B * D * H * W * (C / (dtype == DYCON_BF16 ? 8 : 4));
    DYCON_DISPATCH(dtype, { maxpool2_bwd_kernel<T><<<sgrid(total), 256, 0, stream>>>((const T*)gy, idx, (T*)gx, B, D, H, W, C, total); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_trilinear_fwd(const void* x, void* y, int dtype, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                   int C, int ldy, int coff, int align_corners, dycon_stream_t stream) {
    DYCON_REQUIRE(x && y && B > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0 && C > 0, "trilinear_fwd: bad shape");
    DYCON_REQUIRE(coff >= 0 && coff + C <= ldy, "trilinear_fwd: channel window [%d,%d) outside ld %d", coff, coff + C, ldy);
    const int vn_ = dtype == DYCON_BF16 ? 8 : 4;
    DYCON_REQUIRE(C % vn_ == 0 && ldy % vn_ == 0 && coff % vn_ == 0, "trilinear_fwd: C, ldy, coff must be multiples of %d", vn_);
    const long long total = (long long)B * Do * Ho * Wo * (C / vn_);
    DYCON_DISPATCH(dtype, {
        trilinear_fwd_kernel<T><<<sgrid(total), 256, 0, stream>>>((const T*)x, (T*)y, B, Di, Hi, Wi, Do, Ho, Wo, C, ldy, coff, align_corners, total);
    });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_trilinear_bwd(const void* gy, void* gx, int dtype, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                   int C, int ldy, int coff, int align_corners, dycon_stream_t stream) {
    DYCON_REQUIRE(gy && gx && B > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0 && C > 0, "trilinear_bwd: bad shape");
    DYCON_REQUIRE(coff >= 0 && coff + C <= ldy, "trilinear_bwd: channel window outside ld");
    const int vn_ = dtype == DYCON_BF16 ? 8 : 4;
    DYCON_REQUIRE(C % vn_ == 0 && ldy % vn_ == 0 && coff % vn_ == 0, "trilinear_bwd: C, ldy, coff must be multiples of %d", vn_);
    const long long total = (long long)B * Di * Hi * Wi * (C / vn_);
    DYCON_DISPATCH(dtype, {
        trilinear_bwd_kernel<T><<<sgrid(total), 256, 0, stream>>>((const T*)gy, (T*)gx, B, Di, Hi, Wi, Do, Ho, Wo, C, ldy, coff, align_corners, total);
    });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_copy_channels(const void* src, int lds, int soff, void* dst, int ldd, int doff, long long rows, int C,
                                   int dtype, dycon_stream_t stream) {
    DYCON_REQUIRE(src && dst && rows > 0 && C > 0 && soff >= 0 && doff >= 0 && soff + C <= lds && doff + C <= ldd, "copy_channels: bad arguments");
    DYCON_DISPATCH(dtype, { copy_channels_kernel<T><<<sgrid(rows * C), 256, 0, stream>>>((const T*)src, lds, soff, (T*)dst, ldd, doff, rows, C); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_relu_fwd(const void* z, const void* skip, const float* chan_scale, void* y, int dtype, int B, long long V, int C,
                              dycon_stream_t stream) {
    DYCON_REQUIRE(z && y && B > 0 && V > 0 && C > 0, "relu_fwd: bad arguments");
    const long long total = (long long)B * V * C;
    DYCON_DISPATCH(dtype, { relu_fwd_kernel<T><<<sgrid(total), 256, 0, stream>>>((const T*)z, (const T*)skip, chan_scale, (T*)y, V, C, total); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_relu_bwd(const void* z, const void* gy, const float* chan_scale, void* gz, int dtype, int B, long long V, int C,
                              dycon_stream_t stream) {
    DYCON_REQUIRE(z && gy && gz && B > 0 && V > 0 && C > 0, "relu_bwd: bad arguments");
    const long long total = (long long)B * V * C;
    DYCON_DISPATCH(dtype, { relu_bwd_kernel<T><<<sgrid(total), 256, 0, stream>>>((const T*)z, (const T*)gy, chan_scale, (T*)gz, V, C, total); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_scale_channels(const void* x, const float* scale, void* y, int dtype, int B, long long V, int C,
                                    dycon_stream_t stream) {
    DYCON_REQUIRE(x && scale && y && B > 0 && V > 0 && C > 0, "scale_channels: bad arguments");
    const long long total = (long long)B * V * C;
    DYCON_DISPATCH(dtype, { scale_channels_kernel<T><<<sgrid(total), 256, 0, stream>>>((const T*)x, scale, (T*)y, V, C, total); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_mul_mask(const void* x, const float* mask, float inv_keep, void* y, int dtype, long long n,
                              dycon_stream_t stream) {
    DYCON_REQUIRE(x && mask && y && n > 0, "mul_mask: bad arguments");
    DYCON_DISPATCH(dtype, { mul_mask_kernel<T><<<sgrid(n), 256, 0, stream>>>((const T*)x, mask, inv_keep, (T*)y, n); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_dropout_philox(const void* x, void* y, int dtype, long long n, float p, uint64_t seed, uint64_t offset,
                                    dycon_stream_t stream) {
    DYCON_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "dropout_philox: bad arguments");
    DYCON_DISPATCH(dtype, { dropout_philox_kernel<T><<<sgrid((n + 3) / 4), 256, 0, stream>>>((const T*)x, (T*)y, n, p, seed, offset); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_channel_mask_philox(float* scale, long long n, float p, uint64_t seed, uint64_t offset,
                                         dycon_stream_t stream) {
    DYCON_REQUIRE(scale && n > 0 && p >= 0.f && p < 1.f, "channel_mask_philox: bad arguments");
    channel_mask_philox_kernel<<<cdiv(n, 256), 256, 0, stream>>>(scale, n, p, seed, offset);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_add_noise(const void* x, const float* noise, void* y, int dtype, long long n, float sigma, float clip,
                               uint64_t seed, uint64_t offset, dycon_stream_t stream) {
    DYCON_REQUIRE(x && y && n > 0, "add_noise: bad arguments");
    DYCON_DISPATCH(dtype, { add_noise_kernel<T><<<sgrid((n + 3) / 4), 256, 0, stream>>>((const T*)x, noise, (T*)y, n, sigma, clip, seed, offset); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_tanh(const void* x, int x_dtype, float* y, long long n, dycon_stream_t stream) {
    DYCON_REQUIRE(x && y && n > 0, "tanh: bad arguments");
    DYCON_DISPATCH(x_dtype, { tanh_kernel<T><<<sgrid(n), 256, 0, stream>>>((const T*)x, y, n); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_cast(const void* x, int x_dtype, void* y, int y_dtype, long long n, dycon_stream_t stream) {
    DYCON_REQUIRE(x && y && n > 0, "cast: bad arguments");
    const int g = sgrid(n);
    if (x_dtype == DYCON_F32 && y_dtype == DYCON_BF16) cast_kernel<float, bf16><<<g, 256, 0, stream>>>((const float*)x, (bf16*)y, n);
    else if (x_dtype == DYCON_BF16 && y_dtype == DYCON_F32) cast_kernel<bf16, float><<<g, 256, 0, stream>>>((const bf16*)x, (float*)y, n);
    else if (x_dtype == DYCON_F32 && y_dtype == DYCON_F32) cast_kernel<float, float><<<g, 256, 0, stream>>>((const float*)x, (float*)y, n);
    else if (x_dtype == DYCON_BF16 && y_dtype == DYCON_BF16) cast_kernel<bf16, bf16><<<g, 256, 0, stream>>>((const bf16*)x, (bf16*)y, n);
    else { dycon_set_error("cast: bad dtypes"); return DYCON_ERR_INVALID; }
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_add(const void* a, const void* b, void* y, int dtype, long long n, dycon_stream_t stream) {
    DYCON_REQUIRE(a && y && n > 0, "add: bad arguments");
    DYCON_DISPATCH(dtype, { add_kernel<T><<<sgrid(n), 256, 0, stream>>>((const T*)a, (const T*)b, (T*)y, n); });
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}
