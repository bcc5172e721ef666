// clip_grad_norm_ + SGD(momentum, weight decay) + EMA teacher update over flat fp32 arenas
// (train_DyCON_BraTS19.py:155-164, 268, 369-372).  No host round trip: the clip coefficient is
// computed on the device from the sum of squares, and a non-finite loss (the reference's
// `continue`, :360-362) turns the update into a no-op through a device flag.
#include "common.h"

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, double* __restrict__ out) {
    __shared__ float red[17];
    float acc = 0.f;
    const long long n4 = n / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = g4[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += g[i] * g[i];
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(out, (double)s);
}

__global__ __launch_bounds__(256) void sgd_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom,
                                                      float* __restrict__ teacher, long long n_sgd, long long n_all,
                                                      const double* __restrict__ sumsq, float max_norm, float grad_scale, float lr,
                                                      float momentum, float wd, float alpha, const int* __restrict__ skip) {
    if (skip && *skip) return;
    // gradients in the arena are grad_scale x the true gradient (DDP: sum over ranks, scale = 1/world)
    const float total = sqrtf((float)sumsq[0]) * grad_scale;
    const float coef = fminf(max_norm / (total + 1e-6f), 1.f) * grad_scale;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_all; i += (long long)gridDim.x * 256) {
        float pv = p[i];
        if (i < n_sgd) {
            const float gv = g[i] * coef + wd * pv;
            const float m = momentum * mom[i] + gv;
            mom[i] = m;
            pv -= lr * m;
            p[i] = pv;
        }
        if (teacher) teacher[i] = teacher[i] * alpha + pv * (1.f - alpha);
    }
}

__global__ void nonfinite_flag_kernel(const float* __restrict__ x, int* __restrict__ flag) { flag[0] = isfinite(x[0]) ? 0 : 1; }

__global__ void set_scalars_kernel(float* __restrict__ dst, int n, float v0, float v1, float v2, float v3, float v4, float v5,
                                   float v6, float v7) {
    const float v[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
    if (threadIdx.x < n) dst[threadIdx.x] = v[threadIdx.x];
}

extern "C" int dycon_set_scalars(float* dst, int n, float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                 float v7, dycon_stream_t stream) {
    DYCON_REQUIRE(dst && n > 0 && n <= 8, "set_scalars: bad arguments");
    set_scalars_kernel<<<1, 8, 0, stream>>>(dst, n, v0, v1, v2, v3, v4, v5, v6, v7);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_sumsq(const float* g, long long n, double* sumsq, dycon_stream_t stream) {
    DYCON_REQUIRE(g && sumsq && n > 0, "sumsq: bad arguments");
    DYCON_REQUIRE(((uintptr_t)g & 15) == 0, "sumsq: gradient arena must be 16-byte aligned");
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    sumsq_kernel<<<(int)blocks, 256, 0, stream>>>(g, n, sumsq);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_sgd_ema(float* p, const float* g, float* mom, float* teacher, long long n_sgd, long long n_all,
                             const double* sumsq, float max_norm, float grad_scale, float lr, float momentum, float weight_decay,
                             float ema_alpha, const int* skip_flag, dycon_stream_t stream) {
    DYCON_REQUIRE(p && g && mom && sumsq && n_sgd >= 0 && n_all >= n_sgd && n_all > 0, "sgd_ema: bad arguments");
    long long blocks = (n_all + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    sgd_ema_kernel<<<(int)blocks, 256, 0, stream>>>(p, g, mom, teacher, n_sgd, n_all, sumsq, max_norm, grad_scale, lr, momentum,
                                                    weight_decay, ema_alpha, skip_flag);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_nonfinite_flag(const float* x, int* flag, dycon_stream_t stream) {
    DYCON_REQUIRE(x && flag, "nonfinite_flag: bad arguments");
    nonfinite_flag_kernel<<<1, 1, 0, stream>>>(x, flag);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}
