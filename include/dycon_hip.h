/* libdycon_hip.so -- C ABI of the MI355X-native DyCON training-step kernels.
 *
 * The reference (rogeliorjr/DyCON_Paper_Replication) has no FFI: its hot path is plain
 * Python calling torch.nn / torch.nn.functional.  Each entry point below therefore cites the
 * reference call site (file:line under code/) whose ATen/cuDNN work it replaces; the Python
 * side (dycon_paper_replication_amd/) binds these with ctypes and mirrors the reference's
 * callables (net_factory_3d, UnCLoss, FeCLoss, losses.*, ramps.*).  See INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - activations are channels-last-3D: (B, D, H, W, C) contiguous, C fastest ("NDHWC");
 *   - dtype: DYCON_F32 or DYCON_BF16 storage; accumulation is always fp32;
 *   - outputs and workspaces are caller-allocated, no hidden allocation, no host sync:
 *     every call only enqueues work on `stream` and is hipGraph-capturable;
 *   - return 0 on success, <0 on error; dycon_last_error() gives a thread-local message.
 */
#ifndef DYCON_HIP_H
#define DYCON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* dycon_stream_t; /* == hipStream_t */

#define DYCON_F32 0
#define DYCON_BF16 1

#define DYCON_OK 0
#define DYCON_ERR_INVALID (-22)
#define DYCON_ERR_LAUNCH (-5)

/* gather modes of the convolution family */
#define DYCON_CONV_1X1 0  /* nn.Conv3d(k=1)          VNet.py:175, UNet3D_contrastive.py:249-250,262,265 */
#define DYCON_CONV_K3 1   /* nn.Conv3d(k=3, pad=1)   VNet.py:16, networks/utils.py:104,107              */
#define DYCON_CONV_K2S2 2 /* nn.Conv3d(k=2, stride=2) VNet.py:73                                          */

int dycon_version(void);
const char* dycon_last_error(void);

/* ---------------------------------------------------------------- weight packing
 * Reorders one fp32 weight tensor (torch layout, arbitrary strides) into the MFMA B-fragment
 * order consumed by dycon_conv_gemm:  K = T*Cin (tap-major, channel-minor), GEMM column
 * n = n1*N0 + n0.  Element (t, c, n1, n0) is read from w[t'*s_t + c*s_c + n1*s_n1 + n0*s_n0]
 * with t' = flip_taps ? T-1-t : t (flipped taps give the data-gradient of a k=3 conv).
 * Output bytes: dycon_bfrag_bytes(). */
size_t dycon_bfrag_bytes(int dtype, int T, int Cin, int N);
int dycon_pack_bfrag(const float* w, void* out, int dtype, int T, int Cin, int N, int N0,
                     long long s_t, long long s_c, long long s_n1, long long s_n0, int flip_taps,
                     dycon_stream_t stream);
/* Batched form: every pack of a training step in one launch.  jobs_dev is a DEVICE array of njobs descriptors
 * (built once -- the parameter arenas and packed buffers keep their addresses).  kind: 0 = bf16 fragments,
 * 1 = fp32 fragments, 2 = plain fp32 [T][Cin][N]; total = number of output elements; NT = ceil(N/16). */
typedef struct {
    const float* w;
    void* out;
    long long s_t, s_c, s_n1, s_n0, total;
    int kind, T, Cin, N, N0, flip, NT, pad_;
} dycon_pack_job_t;
int dycon_pack_batch(const dycon_pack_job_t* jobs_dev, int njobs, int blocks_per_job, dycon_stream_t stream);
/* plain fp32 [T][Cin][N] packing for the skinny (Cin<8 or N%16!=0) direct kernels */
int dycon_pack_tcn(const float* w, float* out, int T, int Cin, int N, int N0, long long s_t,
                   long long s_c, long long s_n1, long long s_n0, int flip_taps, dycon_stream_t stream);

/* ---------------------------------------------------------------- convolution family (implicit GEMM on MFMA)
 * y[row, n] (+)= bias[n % Cout] + sum_{t,c} x[src(row,t), c] * W[t, c, n]
 *   rows  = voxels of the output grid (mode 1x1/k3: the input grid; k2s2: the half grid)
 *   scatter=1: N = 8*Cout and column n = tap*Cout + co is written to output voxel
 *              2*row+tap of the doubled grid  == nn.ConvTranspose3d(k=2, s=2)  (VNet.py:100)
 *   accumulate=1: y += result (used to sum the two gradient paths into a skip tensor,
 *              VNet.py:210-222 / networks/utils.py:276).
 * Replaces F.conv3d / F.conv_transpose3d forward and their data-gradients.
 * Needs Cin % 8 == 0 (bf16) or Cin % 4 == 0 (f32), N % 16 == 0; otherwise use *_direct.
 * Exception (bf16, k3, >= 24^3 voxels): Cin == 1 with N in {16, 32, 64} (first layer) runs on the matrix cores
 * with wfrag packed as T=27, Cin=1 (K = 27 taps in one 32-wide k-step); Cin == 48 (U-Net
 * decoder) runs there with wfrag packed chunk-major: three T=27, Cin=16 packs of channels [0,16), [16,32),
 * [32,48) back to back.
 * workspace (optional, dycon_conv_gemm_workspace() bytes): enables split-K for the small spatial
 * levels (6^3, 12^3), whose few row blocks cannot fill 256 CUs: per-split fp32 slabs + ordered finish. */
size_t dycon_conv_gemm_workspace(int dtype, int mode, int scatter, int B, int Di, int Hi, int Wi,
                                 int Cin, int N);
int dycon_conv_gemm(const void* x, const void* wfrag, const float* bias, void* y, int dtype,
                    int mode, int scatter, int accumulate, int B, int Di, int Hi, int Wi, int Cin,
                    int N, int Cout, float* workspace, size_t ws_bytes, dycon_stream_t stream);
/* dycon_conv_gemm with defer_finish: on a split-K shape (dycon_conv_gemm_splits() > 1, workspace given, no accumulation) the
 * partial slabs stay in `workspace` ([split][row][N] fp32) and no finish is launched; the convolution is completed by
 * dycon_norm_fwd_slab, which fuses bias + ordered slab sum + rounding with the GroupNorm / InstanceNorm that follows
 * (VNet.py:16-23 at the 12^3 / 6^3 levels): one launch less on the step's critical chain, bit-identical results. */
int dycon_conv_gemm_splits(int dtype, int mode, int scatter, int B, int Di, int Hi, int Wi, int Cin, int N);
int dycon_conv_gemm_ex(const void* x, const void* wfrag, const float* bias, void* y, int dtype,
                       int mode, int scatter, int accumulate, int B, int Di, int Hi, int Wi, int Cin,
                       int N, int Cout, float* workspace, size_t ws_bytes, int defer_finish,
                       dycon_stream_t stream);
/* skinny channels (first layer 1->16, 1x1 heads 16->2 and their data-gradient 2->16):
 * w_tcn is fp32 [T][Cin][N]; x and y may have different dtypes (bf16 features, fp32 logits). */
int dycon_conv_direct(const void* x, int x_dtype, const float* w_tcn, const float* bias, void* y,
                      int y_dtype, int mode, int accumulate, int B, int Di, int Hi, int Wi, int Cin,
                      int N, dycon_stream_t stream);

/* weight gradient: dw[t*s_t + c*s_c + n*s_n] = sum_rows x[src(row,t), c] * gy[row, n]
 * (x on the input grid with Cin channels, gy on the output-row grid with Cout channels), and,
 * when dbias != NULL, the bias gradient dbias[n] = sum_rows gy[row, n] (fused into the same
 * pass over gy on the bf16 k=3 path).  Two-stage, deterministic: per-split partials in
 * `workspace`, then an ordered reduce. */
size_t dycon_conv_wgrad_workspace(int mode, int B, int Di, int Hi, int Wi, int Cin, int Cout);
int dycon_conv_wgrad(const void* x, int x_dtype, const void* gy, int g_dtype, float* dw, float* dbias,
                     int mode, int B, int Di, int Hi, int Wi, int Cin, int Cout, long long s_t,
                     long long s_c, long long s_n, float* workspace, size_t ws_bytes,
                     dycon_stream_t stream);

/* out[c] = sum_rows x[row, c]   (bias gradients; also per-channel sums) */
size_t dycon_colsum_workspace(long long rows, int C);
int dycon_colsum(const void* x, int dtype, float* out, long long rows, int C, float* workspace,
                 size_t ws_bytes, dycon_stream_t stream);

/* ---------------------------------------------------------------- normalisation (+ReLU, + skip add)
 * One kernel family for nn.GroupNorm(16,C) (VNet.py:20), nn.InstanceNorm3d (G=C, no affine;
 * networks/utils.py:105,108) and train-mode nn.BatchNorm3d (Nb=1, G=C over all B*V rows;
 * UNet3D_contrastive.py:263,266).  x: (Nb, V, C).  stats: (Nb, G, 2) = {mean, rstd}. */
size_t dycon_norm_workspace(int Nb, long long V, int C);
int dycon_norm_stats(const void* x, int dtype, int Nb, long long V, int C, int G, float eps,
                     float* stats, float* running_mean, float* running_var, float momentum,
                     float* workspace, size_t ws_bytes, dycon_stream_t stream);
/* statistics + apply in one call: ONE launch when V <= 2048 (12^3, 6^3 levels) (each workgroup owns whole groups of one sample),
 * otherwise dycon_norm_stats followed by dycon_norm_apply.  Same arguments as those two. */
int dycon_norm_fwd(const void* x, void* y, int dtype, int Nb, long long V, int C, int G, float eps,
                   float* stats, const float* gamma, const float* beta, int relu, const void* skip,
                   const float* chan_scale, float* running_mean, float* running_var, float momentum,
                   float* workspace, size_t ws_bytes, dycon_stream_t stream);
/* 1 when dycon_norm_fwd serves this shape with its one-launch kernel (V <= 2048, whole groups per workgroup) */
int dycon_norm_fwd_is_fused(int dtype, long long V, int C, int G);
/* One-launch norm fed by the split-K slabs of the producing convolution (bf16 storage, shapes with dycon_norm_fwd_is_fused):
 * x_out = bf16(conv_bias + sum_z slab[z]) (kept for the backward), stats, y -- see dycon_conv_gemm_ex. */
int dycon_norm_fwd_slab(const float* slab, int splits, const float* conv_bias, void* x_out, void* y,
                        int dtype, int Nb, long long V, int C, int G, float eps, float* stats,
                        const float* gamma, const float* beta, int relu, const void* skip,
                        const float* chan_scale, dycon_stream_t stream);
/* y = act(gamma*(x-mean)*rstd + beta) * chan_scale[n,c] + skip ; gamma/beta/skip/chan_scale may be
 * NULL; y may alias x.  chan_scale (Nb, C) = keep/(1-p) fuses nn.Dropout3d (VNet.py:177,196,226). */
int dycon_norm_apply(const void* x, void* y, int dtype, int Nb, long long V, int C, int G,
                     const float* stats, const float* gamma, const float* beta, int relu,
                     const void* skip, const float* chan_scale, dycon_stream_t stream);
/* backward.  src = x (from_y=0, xhat=(x-mean)*rstd) or, ONLY when relu=0 (the op is then
 * invertible), y (from_y=1, xhat=(y-beta)/gamma).  gx may alias gy.  dgamma/dbeta may be NULL. */
int dycon_norm_bwd(const void* src, int from_y, const void* gy, void* gx, int dtype, int Nb,
                   long long V, int C, int G, const float* stats, const float* gamma,
                   const float* beta, int relu, const float* chan_scale, float* dgamma, float* dbeta,
                   float* workspace, size_t ws_bytes, dycon_stream_t stream);
/* dycon_norm_bwd with defer_dparams = 1: on the one-launch shapes (dycon_norm_fwd_is_fused) the per-sample {dbeta, dgamma}
 * contributions stay in `workspace` ((Nb, C, 2) floats at its start) and the caller adds them up with dycon_norm_sum_dparams --
 * e.g. on the weight-gradient stream: the parameter gradients only feed the optimiser, the data gradient is the dependent chain.
 * Other shapes ignore the flag (their finalize launch writes dgamma / dbeta). */
int dycon_norm_bwd_ex(const void* src, int from_y, const void* gy, void* gx, int dtype, int Nb,
                      long long V, int C, int G, const float* stats, const float* gamma,
                      const float* beta, int relu, const float* chan_scale, float* dgamma, float* dbeta,
                      int defer_dparams, float* workspace, size_t ws_bytes, dycon_stream_t stream);
int dycon_norm_sum_dparams(const float* workspace, int Nb, int C, float* dgamma, float* dbeta,
                           dycon_stream_t stream);
/* The 2-class head fused into the V-Net's last normalisation (VNet.py:225-227: block_nine's conv -> norm -> ReLU [-> Dropout3d] ->
 * out_conv 1x1; train_DyCON_BraTS19.py:304).  C = 16 channels, 2 logits; head_w = out_conv.weight (2,16,1,1,1) fp32 as torch
 * stores it, head_b = out_conv.bias.  The normalised tensor and its gradient are never written:
 *   fwd : logits (Nb, V, 2) fp32 = head(act(norm(x)) * chan_scale), statistics from dycon_norm_stats
 *   bwd : gx = d/dx given g_logits (Nb, V, 2) fp32; dgamma / dbeta as dycon_norm_bwd; the head's weight / bias gradient partials
 *         stay in `workspace` (dycon_norm_head_workspace bytes) until dycon_norm_head_dparams sums them (any stream).
 * Per-voxel arithmetic and roundings are those of dycon_norm_apply + dycon_conv_direct (forward) and dycon_conv_direct's data
 * gradient + dycon_norm_bwd (backward): same logits bit for bit, same gx to fp32 round-off. */
size_t dycon_norm_head_workspace(int Nb, long long V);
int dycon_norm_head_fwd(const void* x, int dtype, int Nb, long long V, int G, const float* stats, const float* gamma,
                        const float* beta, int relu, const float* chan_scale, const float* head_w, const float* head_b,
                        float* logits, dycon_stream_t stream);
int dycon_norm_head_bwd(const void* x, const float* g_logits, void* gx, int dtype, int Nb, long long V, int G,
                        const float* stats, const float* gamma, const float* beta, int relu, const float* chan_scale,
                        const float* head_w, float* dgamma, float* dbeta, float* workspace, size_t ws_bytes,
                        dycon_stream_t stream);
int dycon_norm_head_dparams(const float* workspace, int Nb, long long V, float* d_head_w, float* d_head_b,
                            dycon_stream_t stream);
/* The normalisation after the FIRST convolution (block_one, VNet.py:176) has one consumer of its data gradient: that convolution's
 * weight gradient (the image needs none).  So the gradient is never stored:
 *   dycon_norm_bwd_stats     = the first two launches of dycon_norm_bwd (statistics pass + finalize): dgamma / dbeta, and the per-group
 *                              {A, B} sums at float offset dycon_norm_bwd_ab_offset(Nb, V, C) of `workspace` (dycon_norm_workspace bytes);
 *   dycon_conv1_wgrad_normbwd = weight + bias gradient of a 1 -> 16 channel k=3 convolution (bf16) reading x (the image), gy (gradient
 *                              w.r.t. the normalisation's output) and z (its input); gz = norm-backward(gy, z) is formed per element on
 *                              load, rounded to bf16 as the stored tensor would have been.  dw layout as dycon_conv_wgrad.
 * Replaces the backward-apply pass (2 reads + 1 write of the step's largest tensor) and the weight gradient's read of its result. */
/* Statistics of a normalisation's input taken by the convolution that produces it.  On the shapes dycon_conv_stats_chunks serves (bf16
 * k=3 32 -> 32 on >= 1024 tiles: the persistent kernel of the 48^3 level) dycon_conv_gemm_stats is dycon_conv_gemm that also leaves
 * per-(sample, chunk, channel) {sum, sum of squares} of the values it STORED in stat_part ([B][chunks][Cout][2] floats); the
 * normalisation that follows is then dycon_norm_fwd_parts = finalize + apply, without the pass that re-reads the tensor
 * (BatchNorm: Nb = 1, chunks = B * chunks). */
int dycon_conv_stats_chunks(int dtype, int mode, int B, int Di, int Hi, int Wi, int Cin, int Cout);
/* (also served since round 3: the persistent kernels of the 96^3 level, bf16 k=3 16 -> 16 and 1 -> 16.)  dycon_norm_stats_parts is the
 * finalize alone (stats = mean / rstd per (n, g)), for consumers that apply the statistics themselves (dycon_norm_head_fwd). */
int dycon_norm_stats_parts(int dtype, int Nb, long long V, int C, int G, float eps, float* stats, float* running_mean,
                           float* running_var, float momentum, const float* part, int chunks, dycon_stream_t stream);
int dycon_conv_gemm_stats(const void* x, const void* wfrag, const float* bias, void* y, int dtype, int B, int Di, int Hi,
                          int Wi, int Cin, int Cout, float* stat_part, size_t stat_bytes, dycon_stream_t stream);
int dycon_norm_fwd_parts(const void* x, void* y, int dtype, int Nb, long long V, int C, int G, float eps, float* stats,
                         const float* gamma, const float* beta, int relu, const void* skip, const float* chan_scale,
                         float* running_mean, float* running_var, float momentum, const float* part, int chunks,
                         dycon_stream_t stream);
size_t dycon_norm_bwd_ab_offset(int Nb, long long V, int C);
int dycon_norm_bwd_stats(const void* src, const void* gy, int dtype, int Nb, long long V, int C, int G, const float* stats,
                         const float* gamma, const float* beta, int relu, const float* chan_scale, float* dgamma,
                         float* dbeta, float* workspace, size_t ws_bytes, dycon_stream_t stream);
/* The whole backward of block_one in ONE pass over (x, z, gy): the normalisation's data gradient is affine in (g', z) with
 * coefficients that depend on group sums the same pass takes, so the first convolution's weight gradient is a combination of three tap
 * correlations (sum g' x, sum z x, sum x) that do not depend on them (csrc/conv.hip, first_block_bwd_kernel).  Writes dw / dbias of the
 * convolution (layout as dycon_conv_wgrad) and dgamma / dbeta of the normalisation; no bf16 rounding of the data gradient enters dw.
 * Replaces dycon_norm_bwd_stats + dycon_conv1_wgrad_normbwd (the inputs are read once instead of twice). */
size_t dycon_first_block_bwd_workspace(int B, int D, int H, int W);
int dycon_first_block_bwd(const void* x, const void* z, const void* gy, int B, int D, int H, int W, int Nb, int G,
                          const float* stats, const float* gamma, const float* beta, int relu, const float* chan_scale,
                          float* dgamma, float* dbeta, float* dw, float* dbias, long long s_t, long long s_c, long long s_n,
                          float* workspace, size_t ws_bytes, dycon_stream_t stream);
size_t dycon_conv1_wgrad_normbwd_workspace(int B, int D, int H, int W);
int dycon_conv1_wgrad_normbwd(const void* x, const void* z, const void* gy, int B, int D, int H, int W, int Nb, int G,
                              const float* stats, const float* gamma, const float* beta, int relu, const float* chan_scale,
                              const float* ab, float* dw, float* dbias, long long s_t, long long s_c, long long s_n,
                              float* workspace, size_t ws_bytes, dycon_stream_t stream);
/* Accumulator forms of dycon_norm_fwd / dycon_norm_bwd (from_y = 0): `acc` = dycon_norm_acc_doubles(Nb, V, C) doubles that are
 * ZERO on entry (a slice of an arena the caller clears once per step).  On the shapes that are not served by the one-launch kernels, every chunk of the
 * statistics pass adds its sums to acc (double atomics) and the apply pass forms the group statistics in its prologue: TWO launches
 * instead of three, forward (stats written for the backward, BatchNorm running statistics updated) and backward (dgamma / dbeta
 * written by the apply pass).  Same results up to the last bits of the double sums.  workspace: only used on the one-launch shapes. */
size_t dycon_norm_acc_doubles(int Nb, long long V, int C);
int dycon_norm_fwd_acc(const void* x, void* y, int dtype, int Nb, long long V, int C, int G, float eps,
                       float* stats, const float* gamma, const float* beta, int relu, const void* skip,
                       const float* chan_scale, float* running_mean, float* running_var, float momentum,
                       double* acc, dycon_stream_t stream);
int dycon_norm_bwd_acc(const void* src, const void* gy, void* gx, int dtype, int Nb, long long V, int C,
                       int G, const float* stats, const float* gamma, const float* beta, int relu,
                       const float* chan_scale, float* dgamma, float* dbeta, double* acc, float* workspace,
                       size_t ws_bytes, dycon_stream_t stream);

/* ---------------------------------------------------------------- data movement / pointwise
 * nn.MaxPool3d(2) (UNet3D_contrastive.py:225-237); idx holds the first-max position 0..7 */
int dycon_maxpool2_fwd(const void* x, void* y, uint8_t* idx, int dtype, int B, int D, int H, int W,
                       int C, dycon_stream_t stream);
int dycon_maxpool2_bwd(const void* gy, const uint8_t* idx, void* gx, int dtype, int B, int D, int H,
                       int W, int C, dycon_stream_t stream);
/* trilinear resize (networks/utils.py:264 align_corners=False; UNet3D_contrastive.py:309 True).
 * y/gy rows have `ldy` channels per voxel and the op touches channels [coff, coff+C): this is how
 * the decoder's torch.cat([skip, up]) (networks/utils.py:276) is written without a copy pass. */
int dycon_trilinear_fwd(const void* x, void* y, int dtype, int B, int Di, int Hi, int Wi, int Do,
                        int Ho, int Wo, int C, int ldy, int coff, int align_corners,
                        dycon_stream_t stream);
int dycon_trilinear_bwd(const void* gy, void* gx, int dtype, int B, int Di, int Hi, int Wi, int Do,
                        int Ho, int Wo, int C, int ldy, int coff, int align_corners,
                        dycon_stream_t stream);
/* dst[row, doff + c] = src[row, soff + c]  (concat / split of channel slices) */
int dycon_copy_channels(const void* src, int lds, int soff, void* dst, int ldd, int doff,
                        long long rows, int C, int dtype, dycon_stream_t stream);
/* ReLU of the V-Net's normalization='none' blocks (VNet.py:23-24, 87, 114), with the Dropout3d channel factor and the decoder's
 * skip add folded in like in the norm kernels: y = relu(z) * chan_scale[b, c] + skip; gz = (z > 0) * gy * chan_scale[b, c].
 * skip / chan_scale may be NULL. */
int dycon_relu_fwd(const void* z, const void* skip, const float* chan_scale, void* y, int dtype, int B, long long V, int C,
                   dycon_stream_t stream);
int dycon_relu_bwd(const void* z, const void* gy, const float* chan_scale, void* gz, int dtype, int B, long long V, int C,
                   dycon_stream_t stream);

/* y[b, v, c] = x[b, v, c] * scale[b, c]   (nn.Dropout3d(0.5), VNet.py:177,196,226; fwd and bwd) */
int dycon_scale_channels(const void* x, const float* scale, void* y, int dtype, int B, long long V,
                         int C, dycon_stream_t stream);
/* y = x * mask * inv_keep  with an explicit float keep-mask (parity tests) */
int dycon_mul_mask(const void* x, const float* mask, float inv_keep, void* y, int dtype,
                   long long n, dycon_stream_t stream);
/* Philox4x32-10 element-wise dropout (nn.Dropout(0.3), UNet3D_contrastive.py:253-254):
 * keep = u > p ; the same (seed, offset) regenerates the mask in backward */
int dycon_dropout_philox(const void* x, void* y, int dtype, long long n, float p, uint64_t seed,
                         uint64_t offset, dycon_stream_t stream);
/* scale[i] = bernoulli(1-p)/(1-p)  for i < n   (channel masks of Dropout3d) */
int dycon_channel_mask_philox(float* scale, long long n, float p, uint64_t seed, uint64_t offset,
                              dycon_stream_t stream);
/* y = x + clamp(N(0,1)*sigma, -clip, clip)   (train_DyCON_BraTS19.py:301-302); noise==NULL:
 * Philox + Box-Muller, else the explicit noise tensor is added (parity tests) */
int dycon_add_noise(const void* x, const float* noise, void* y, int dtype, long long n, float sigma,
                    float clip, uint64_t seed, uint64_t offset, dycon_stream_t stream);
int dycon_tanh(const void* x, int x_dtype, float* y, long long n, dycon_stream_t stream);
int dycon_cast(const void* x, int x_dtype, void* y, int y_dtype, long long n, dycon_stream_t stream);
/* y = a + b (b may be NULL -> copy) */
int dycon_add(const void* a, const void* b, void* y, int dtype, long long n, dycon_stream_t stream);

/* ---------------------------------------------------------------- voxel losses (2 classes)
 * One pass over student+teacher logits (fp32, (B,V,2)) and labels accumulates every sum the
 * step needs (train_DyCON_BraTS19.py:308-314,351-352; utils/losses.py:8-16,65-104,156-192;
 * utils/dycon_losses.py:94-118) into sums[16] (double, zeroed by the call):
 *   0 ce_sum   1 I1  2 Z1  3 Y1   4 I0  5 Z0  6 Y0   7 mse_sum  8 kl_sum  9 uncl_w  10 uncl_h
 * samples [0,LB) are labelled (CE, Dice), [LB,B) feed the consistency term, all feed UnCL.
 * fast != 0: hardware exp2 / log2 / rcp sequences (1-2 ulp) instead of the libm routines -- for
 * logits that come out of a bf16 network; the fp32 parity mode passes 0. */
int dycon_seg_losses_fwd(const float* s_logits, const float* t_logits, const void* labels,
                         int label_bytes, int B, int LB, long long V, float beta, double* sums,
                         int fast, dycon_stream_t stream);
/* g_logits = sum_k coef[k] * d(loss_k)/d(s_logits); coef (device, 5 floats): ce, dice_fg,
 * dice_multiclass, consistency (mse or kl per cons_kind), uncl -- already multiplied by the
 * upstream gradient.  Needs the sums from the forward call. */
int dycon_seg_losses_bwd(const float* s_logits, const float* t_logits, const void* labels,
                         int label_bytes, int B, int LB, long long V, float beta, const double* sums,
                         const float* coef, int cons_kind, float* g_logits, int fast,
                         dycon_stream_t stream);

/* vals[6] (device floats) = ce, dice(class 1), dice(mean over classes), cons mse, cons kl, uncl */
int dycon_seg_losses_finalize(const double* sums, int B, int LB, long long V, float beta, float* vals,
                              dycon_stream_t stream);
/* out[6] = total, ce, dice, cons, fecl, uncl with total = l_w*(ce+dice) + cons_w*cons + u_w*(fecl+uncl)
 * (train_DyCON_BraTS19.py:355-357); nonfinite[0] = !isfinite(total) (the NaN/Inf guard, :360-362).
 * dice_kind 0: class-1 dice, 1: multi-class; cons_kind 0: mse, 1: kl; fecl may be NULL. */
int dycon_step_loss(const float* vals, const float* fecl, float l_weight, float cons_weight,
                    float u_weight, int dice_kind, int cons_kind, float* out, int* nonfinite,
                    dycon_stream_t stream);
/* The same two reductions and the FeCL finalize (dycon_fecl_finalize) as ONE launch: the scalar end of the step's loss forward
 * from the raw accumulators -- sums (16 doubles of dycon_seg_losses_fwd) and fecl_out (4 doubles of dycon_fecl_fwd, may be NULL),
 * after a data-parallel run has all-reduced them (B, LB, fecl_rows are then the GLOBAL counts).  out[6] as dycon_step_loss. */
int dycon_step_losses(const double* sums, const double* fecl_out, int B, int LB, long long V, float beta,
                      double fecl_rows, float lambda_cross, int has_teacher, float l_weight, float cons_weight,
                      float u_weight, int dice_kind, int cons_kind, float* out, int* nonfinite,
                      dycon_stream_t stream);

/* ---------------------------------------------------------------- the reference's loss callables, reference semantics
 * The fused pass above takes LOGITS; the reference's own step body instead calls utils/losses.py on PROBABILITIES it computed
 * with torch (train_DyCON_BraTS19.py:308-314,352).  These entry points implement those callables literally, so that loop body
 * runs unchanged on this library.  All tensors are fp32 and described by a strided (n, C, V) view (element strides): channel
 * stride 1 for channels-last maps, V for plain NCDHW, element stride 2 for `probs[:, 1]`.  1 <= C <= 8. */
typedef struct {
    const void* p;          /* device pointer of element (0, 0, 0) */
    long long sn, sc, sv;   /* element strides of the sample, channel and (flattened) voxel index */
} dycon_view_t;
/* losses.softmax_mse_loss(a, b, sigmoid) (utils/losses.py:65-82): out = (softmax(a,1) - softmax(b,1))^2, element-wise */
int dycon_softmax_mse_fwd(const dycon_view_t* a, const dycon_view_t* b, const dycon_view_t* out, long long n, int C,
                          long long V, int sigmoid, dycon_stream_t stream);
/* ga = d sum(g * out) / d a  (the loss is symmetric: pass (b, a) for the gradient of the second argument) */
int dycon_softmax_mse_bwd(const dycon_view_t* a, const dycon_view_t* b, const dycon_view_t* g, const dycon_view_t* ga,
                          long long n, int C, long long V, int sigmoid, dycon_stream_t stream);
/* losses.softmax_kl_loss(a, b, sigmoid) (utils/losses.py:85-104) = F.kl_div(log_softmax(a,1), softmax(b,1), 'mean'):
 * out[0] = sum q (log q - log p) / (n C V); sum: one double of scratch (zeroed by the call). */
int dycon_softmax_kl_fwd(const dycon_view_t* a, const dycon_view_t* b, long long n, int C, long long V, int sigmoid,
                         double* sum, float* out, dycon_stream_t stream);
/* grad = g_up[0] * d out / d (which == 0 ? a : b) */
int dycon_softmax_kl_bwd(const dycon_view_t* a, const dycon_view_t* b, long long n, int C, long long V, int sigmoid,
                         int which, const float* g_up, const dycon_view_t* grad, dycon_stream_t stream);
/* losses.dice_loss(score, target) (utils/losses.py:8-16; C = 1, onehot = 0, target of the score's shape) and
 * losses.DiceLoss(C)(inputs, target, weight, softmax) (utils/losses.py:156-192; onehot = 1: target is the (n, V) label map,
 * class c's target is label == c):  out[0] = sum_c w_c (1 - (2 I_c + 1e-5)/(Z_c + Y_c + 1e-5)) / n_div, with the sums taken
 * over the WHOLE view (batch-global, as the reference).  target_kind: 0 float32, 1 one byte (uint8 / bool), 2 int64.
 * weights_host: C host floats or NULL (= 1).  sums: 24 doubles of scratch, kept for the backward. */
int dycon_dice_fwd(const dycon_view_t* score, const dycon_view_t* target, int target_kind, int onehot, long long n, int C,
                   long long V, int softmax, const float* weights_host, float n_div, double* sums, float* out,
                   dycon_stream_t stream);
int dycon_dice_bwd(const dycon_view_t* score, const dycon_view_t* target, int target_kind, int onehot, long long n, int C,
                   long long V, int softmax, const float* weights_host, float n_div, const double* sums, const float* g_up,
                   const dycon_view_t* grad, dycon_stream_t stream);

/* rows of (R, C): y = x / max(||x||, eps)   (F.normalize, train_DyCON_BraTS19.py:316-323) */
int dycon_l2norm_fwd(const void* x, void* y, float* norms, int dtype, long long R, int C, float eps,
                     dycon_stream_t stream);
int dycon_l2norm_bwd(const void* y, const float* norms, const void* gy, void* gx, int dtype,
                     long long R, int C, float eps, dycon_stream_t stream);
/* mask[b, n] = avg_pool3d(label, k) > 0.5   (train_DyCON_BraTS19.py:326-330) */
int dycon_mask_pool(const void* labels, int label_bytes, float* mask, int B, int D, int H, int W,
                    int kd, int kh, int kw, dycon_stream_t stream);

/* ---------------------------------------------------------------- FeCL (utils/dycon_losses.py:150-235)
 * Blockwise: the (B,N,N) similarity matrix is never materialised; Gram tiles are recomputed on
 * MFMA in each of the passes.  Columns are split over workgroups (partial row results in slabs,
 * combined in order on load): deterministic, and enough workgroups to fill the chip at N = 1728.  feat/teacher: (B,N,Dm) L2-normalised rows; mask: (B,N) float.
 * out (device, 4 doubles, zeroed by the call): student_sum, cross_num, cross_cnt, unused.
 * loss[0] (device float) = student_sum/(B*N) + lambda_cross*cross_num/(cross_cnt+1e-18). */
size_t dycon_fecl_workspace(int B, int N, int Dm);
int dycon_fecl_fwd(const void* feat, const void* teacher, const float* mask, const float* gambling,
                   int dtype, int B, int N, int Dm, float temperature, float gamma, int use_focal,
                   float cross_thresh, float lambda_cross, double* out, float* loss, float* workspace,
                   size_t ws_bytes, dycon_stream_t stream);
/* recompute loss[0] from `out` (after the sums were all-reduced across ranks); rows = global B*N */
int dycon_fecl_finalize(const double* out, double rows, float lambda_cross, int has_teacher, float* loss,
                        dycon_stream_t stream);
/* g_feat = coef[0] * d(loss)/d(feat); needs workspace and `out` untouched since the forward */
int dycon_fecl_bwd(const void* feat, const void* teacher, const float* mask, const float* gambling,
                   int dtype, int B, int N, int Dm, float temperature, float gamma, int use_focal,
                   float cross_thresh, float lambda_cross, const double* out, const float* coef,
                   void* g_feat, float* workspace, size_t ws_bytes, dycon_stream_t stream);

/* ---------------------------------------------------------------- optimiser (train_DyCON_BraTS19.py:155-164,268,369-372)
 * sumsq[0] += sum g^2 (double, caller zeroes) -> clip_grad_norm_; then one fused pass:
 *   coef = min(1, max_norm/(sqrt(sumsq)+1e-6)); g = coef*g + wd*p; m = mu*m + g; p -= lr*m   on [0,n_sgd)
 *   teacher = alpha*teacher + (1-alpha)*p                                                    on [0,n_all)
 * skip_flag (device int, may be NULL): non-zero -> leave everything untouched (NaN/Inf loss). */
int dycon_sumsq(const float* g, long long n, double* sumsq, dycon_stream_t stream);
int dycon_sgd_ema(float* p, const float* g, float* mom, float* teacher, long long n_sgd,
                  long long n_all, const double* sumsq, float max_norm, float grad_scale, float lr,
                  float momentum, float weight_decay, float ema_alpha, const int* skip_flag,
                  dycon_stream_t stream);
/* dst[i] = v_i for i < n <= 8: host scalars (loss weights) into device memory without a copy engine round trip */
int dycon_set_scalars(float* dst, int n, float v0, float v1, float v2, float v3, float v4, float v5,
                      float v6, float v7, dycon_stream_t stream);
/* flag[0] = !isfinite(x[0]) */
int dycon_nonfinite_flag(const float* x, int* flag, dycon_stream_t stream);

/* ---------------------------------------------------------------- sliding-window evaluation (code/utils/test_3d_patch.py:293-351)
 * score / cnt: fp32 maps of the (padded) volume D0 x D1 x D2, zeroed by the caller.  After a batch of n_patches <= 64 windows
 * went through the network, add for every voxel the class-1 softmax of each window covering it (logits: (n_patches, p0, p1, p2, 2)
 * fp32; origins_dev: n_patches x 3 ints, window corner in the volume) and the window count.  Voxel-centric, fixed window order:
 * deterministic although windows overlap. */
int dycon_sw_accumulate(const float* logits, int n_patches, int p0, int p1, int p2, const int* origins_dev,
                        float* score, float* cnt, int D0, int D1, int D2, dycon_stream_t stream);
/* prob = score / cnt (may be NULL), label = prob > thresh   (:344-345) */
int dycon_sw_finalize(const float* score, const float* cnt, long long n, float thresh, uint8_t* label,
                      float* prob, dycon_stream_t stream);
/* out3 (device, caller zeroes) += { |pred|, |gt|, |pred & gt| }: the counts behind medpy.metric.binary.dc / jc
 * (test_3d_patch.py:496-508).  gt: uint8 (gt_bytes 1) or int64 (8), non-zero = foreground. */
int dycon_binary_overlap(const uint8_t* pred, const void* gt, int gt_bytes, long long n,
                         unsigned long long* out3, dycon_stream_t stream);

/* Train-time batch metrics (train_DyCON_BraTS19.py:385-392; utils/metrics.py compute_dice / compute_jaccard): for each sample b
 * out3b[3b..3b+2] (device, caller zeroes) += { |pred|, |gt|, |pred & gt| } with pred = softmax(logits)[1] > 0.5, read straight from
 * the (B, V, 2) fp32 logits. */
int dycon_batch_overlap(const float* logits, const void* gt, int gt_bytes, int B, long long V,
                        unsigned long long* out3b, dycon_stream_t stream);

/* ---------------------------------------------------------------- per-kernel timing (bench.py `roofline`; diagnostics)
 * dycon_kernel_timing(1): from now on every kernel this library launches is bracketed by two HIP timing events on its launch
 * stream (earlier records are dropped); (0): stop.  Each LAUNCH is one record -- the finalize / reduce launch an entry point
 * enqueues after its main kernel is a record of its own.  _count: launches recorded so far (callers bracket an entry point by
 * reading it before and after the call).  _fetch: durations (ms), launch streams and kernel-name ids of records [first, first+n);
 * waits for those launches.  _name: the demangled kernel name of an id.  Not thread-safe; timing runs are single-threaded. */
int dycon_kernel_timing(int on);
long long dycon_kernel_timing_count(void);
int dycon_kernel_timing_fetch(long long first, long long n, float* ms, unsigned long long* stream, int* name_id);
const char* dycon_kernel_timing_name(int id);

#ifdef __cplusplus
}
#endif
#endif /* DYCON_HIP_H */
