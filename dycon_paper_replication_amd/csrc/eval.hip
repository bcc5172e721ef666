// Sliding-window evaluation on the device: code/utils/test_3d_patch.py:293-351 (test_single_case) and the
// overlap counts behind medpy's dc / jc (code/utils/test_3d_patch.py:496-508, calculate_metric_percase).
//
// The reference copies every patch's softmax to the host and accumulates score / count maps with numpy slicing.  Here the maps
// stay in HBM: after a batch of patches went through the network, one voxel-centric pass adds, for every volume voxel, the class-1
// probability of each patch of the batch that covers it (fixed patch order: deterministic, no atomics although windows overlap).
#include "common.h"

__global__ __launch_bounds__(256) void sw_accumulate_kernel(const float* __restrict__ logits, int nb, int p0, int p1, int p2,
                                                            const int* __restrict__ origins, float* __restrict__ score,
                                                            float* __restrict__ cnt, int D0, int D1, int D2) {
    __shared__ int org[3 * 64];
    for (int i = threadIdx.x; i < 3 * nb; i += 256) org[i] = origins[i];
    __syncthreads();
    const long long total = (long long)D0 * D1 * D2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % D2);
        const long long q = i / D2;
        const int b_ = (int)(q % D1), a = (int)(q / D1);
        float s = 0.f, n = 0.f;
        for (int p = 0; p < nb; ++p) {
            const int la = a - org[3 * p], lb = b_ - org[3 * p + 1], lc = c - org[3 * p + 2];
            if ((unsigned)la < (unsigned)p0 && (unsigned)lb < (unsigned)p1 && (unsigned)lc < (unsigned)p2) {
                const float2 l = *reinterpret_cast<const float2*>(logits + ((((long long)p * p0 + la) * p1 + lb) * p2 + lc) * 2);
                s += 1.f / (1.f + __expf(l.x - l.y));        // softmax(logits)[1]   (test_3d_patch.py:334-337)
                n += 1.f;
            }
        }
        if (n > 0.f) { score[i] += s; cnt[i] += n; }
    }
}

__global__ __launch_bounds__(256) void sw_finalize_kernel(const float* __restrict__ score, const float* __restrict__ cnt, long long n,
                                                          float thresh, unsigned char* __restrict__ label, float* __restrict__ prob) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float p = score[i] / cnt[i];                   // every voxel is covered by at least one window
        if (prob) prob[i] = p;
        label[i] = p > thresh ? 1 : 0;                       // (score_map[0] > 0.5), test_3d_patch.py:345
    }
}

template <typename TG>
__global__ __launch_bounds__(256) void binary_overlap_kernel(const unsigned char* __restrict__ pred, const TG* __restrict__ gt, long long n,
                                                             unsigned long long* __restrict__ out) {
    __shared__ unsigned long long sm[3][4];
    unsigned long long a = 0, b = 0, c = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const bool p = pred[i] != 0, g = gt[i] != 0;
        a += p;
        b += g;
        c += p && g;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sm[0][w] = a; sm[1][w] = b; sm[2][w] = c; }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(out + threadIdx.x, sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3]);
}

// Train-time batch metrics (code/train_DyCON_BraTS19.py:385-392): per sample {|pred|, |gt|, |pred & gt|} with pred = softmax(logits)[1] > 0.5
// (<=> logit 1 > logit 0), straight from the student logits -- the reference materialises the probability and binary volumes first.
template <typename TG>
__global__ __launch_bounds__(256) void batch_overlap_kernel(const float2* __restrict__ logits, const TG* __restrict__ gt, long long V,
                                                            unsigned long long* __restrict__ out) {
    __shared__ unsigned long long sm[3][4];
    const int b = blockIdx.y;
    unsigned long long a = 0, g_ = 0, c = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long long)gridDim.x * 256) {
        const float2 l = logits[(long long)b * V + i];
        const bool p = l.y > l.x, g = gt[(long long)b * V + i] != 0;
        a += p;
        g_ += g;
        c += p && g;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); g_ += __shfl_xor(g_, o, 64); c += __shfl_xor(c, o, 64); }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sm[0][w] = a; sm[1][w] = g_; sm[2][w] = c; }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(out + 3 * b + threadIdx.x, sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3]);
}

extern "C" int dycon_batch_overlap(const float* logits, const void* gt, int gt_bytes, int B, long long V, unsigned long long* out3b,
                                   dycon_stream_t stream) {
    DYCON_REQUIRE(logits && gt && out3b && B > 0 && B <= 65535 && V > 0, "batch_overlap: bad arguments");
    DYCON_REQUIRE(gt_bytes == 1 || gt_bytes == 8, "batch_overlap: ground truth must be uint8 or int64");
    long long blocks = (V + 255) / 256;
    if (blocks > 512) blocks = 512;
    dim3 grid((int)blocks, B);
    if (gt_bytes == 1) batch_overlap_kernel<unsigned char><<<grid, 256, 0, stream>>>((const float2*)logits, (const unsigned char*)gt, V, out3b);
    else batch_overlap_kernel<long long><<<grid, 256, 0, stream>>>((const float2*)logits, (const long long*)gt, V, out3b);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_sw_accumulate(const float* logits, int n_patches, int p0, int p1, int p2, const int* origins_dev, float* score,
                                   float* cnt, int D0, int D1, int D2, dycon_stream_t stream) {
    DYCON_REQUIRE(logits && origins_dev && score && cnt, "sw_accumulate: null pointer");
    DYCON_REQUIRE(n_patches > 0 && n_patches <= 64 && p0 > 0 && p1 > 0 && p2 > 0 && D0 >= p0 && D1 >= p1 && D2 >= p2,
                  "sw_accumulate: bad shape (%d patches of %dx%dx%d in %dx%dx%d)", n_patches, p0, p1, p2, D0, D1, D2);
    long long blocks = ((long long)D0 * D1 * D2 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    sw_accumulate_kernel<<<(int)blocks, 256, 0, stream>>>(logits, n_patches, p0, p1, p2, origins_dev, score, cnt, D0, D1, D2);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_sw_finalize(const float* score, const float* cnt, long long n, float thresh, uint8_t* label, float* prob,
                                 dycon_stream_t stream) {
    DYCON_REQUIRE(score && cnt && label && n > 0, "sw_finalize: bad arguments");
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    sw_finalize_kernel<<<(int)blocks, 256, 0, stream>>>(score, cnt, n, thresh, label, prob);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}

extern "C" int dycon_binary_overlap(const uint8_t* pred, const void* gt, int gt_bytes, long long n, unsigned long long* out3,
                                    dycon_stream_t stream) {
    DYCON_REQUIRE(pred && gt && out3 && n > 0, "binary_overlap: bad arguments");
    DYCON_REQUIRE(gt_bytes == 1 || gt_bytes == 8, "binary_overlap: ground truth must be uint8 or int64");
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (gt_bytes == 1) binary_overlap_kernel<unsigned char><<<(int)blocks, 256, 0, stream>>>(pred, (const unsigned char*)gt, n, out3);
    else binary_overlap_kernel<long long><<<(int)blocks, 256, 0, stream>>>(pred, (const long long*)gt, n, out3);
    DYCON_LAUNCH_CHECK();
    return DYCON_OK;
}
