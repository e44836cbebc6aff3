/*
 * sis_hip.h -- C ABI of libsis_hip.so: the MI355X (gfx950) kernels behind the
 * synthesis-in-style hot path (StyleGAN2 generator forward + segmentation training step).
 *
 * Drop-in boundary.  Every entry point takes plain device pointers, sizes and a HIP stream
 * (passed as void*, i.e. a hipStream_t; NULL = the legacy default stream).  No torch types.
 * Inputs are borrowed and must be contiguous; outputs are caller-allocated.  Nothing here
 * synchronises the host.  Return value: 0 on success, non-zero on error (message from
 * sis_last_error(), thread-local).  The reference interface each function replaces is cited
 * as file:line relative to /root/reference/stylegan_code_finder/.
 *
 * dtype codes: 0 = float32, 1 = float64, 2 = float16, 3 = bfloat16.
 */
#ifndef SIS_HIP_H
#define SIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIS_F32 0
#define SIS_F64 1
#define SIS_F16 2
#define SIS_BF16 3

/* library version (major*1000 + minor) and last error text of the calling thread */
int sis_version(void);
const char* sis_last_error(void);
/* Name of the device kernel the calling thread's last sis_modconv2d / sis_modconv2d_up call dispatched to (the
 * dispatch depends on shape and alignment); bench.py uses it so that its per-kernel timings carry the names
 * rocprofv3 reports. */
const char* sis_last_kernel(void);

/* ------------------------------------------------------------------------------------------
 * K1  fused bias + activation.
 * Replaces the pybind entry `fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)`
 *   networks/stylegan2/op/fused_bias_act.cpp:11-21, launcher fused_bias_act_kernel.cu:52-99,
 *   element formula fused_bias_act_kernel.cu:25-47.
 * out[i] = f(x[i] + bias[(i / step_b) % size_b]) * scale, mode = act*10+grad:
 *   10/11 linear; 30 leaky-relu(alpha); 31 leaky-relu gradient gated on ref[i] > 0; 12/32 zero.
 * bias == NULL (or size_b == 0) means "no bias", ref == NULL means "no ref" (the reference
 * passes empty tensors, fused_bias_act_kernel.cu:62-63).  step_b = product of dims[2:].
 */
int sis_fused_bias_act(void* out, const void* x, const void* bias, const void* ref, int dtype,
                       int64_t numel, int64_t step_b, int64_t size_b, int act, int grad,
                       float alpha, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2  upfirdn2d: zero-insert upsample, pad/crop, 2-D FIR with the flipped taps, decimate.
 * Replaces the pybind entry `upfirdn2d_op.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y,
 *   pad_x0, pad_x1, pad_y0, pad_y1)`  networks/stylegan2/op/upfirdn2d.cpp:12-23,
 *   launcher/dispatch upfirdn2d_kernel.cu:140-272, kernel :52-137.
 * in  [major, in_h, in_w, minor], taps [kh, kw] (same dtype as in), out [major, out_h, out_w, minor]
 * with out = (in*up + pad0 + pad1 - k + down) / down  (upfirdn2d_kernel.cu:167-168).
 * Unlike the reference (which silently returns uninitialised memory for an up/down/tap
 * combination outside its 6 modes, upfirdn2d_kernel.cu:172-175) every combination is computed.
 */
int sis_upfirdn2d_out_size(int in_size, int up, int down, int pad0, int pad1, int k);
int sis_upfirdn2d(void* out, const void* in, const void* taps, int dtype, int major, int in_h,
                  int in_w, int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                  int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* ------------------------------------------------------------------------------------------
 * Generator building blocks (float32).  Together they replace the composite PyTorch code of
 * networks/stylegan2/model.py cited per function.
 */

/* PixelNorm, model.py:19-20: out[b,:] = x[b,:] * rsqrt(mean(x[b,:]^2) + 1e-8). */
int sis_pixel_norm(float* out, const float* x, int batch, int dim, void* stream);

/* Output features per workgroup of the three head GEMMs below (EqualLinear, batched modulation, batched demodulation:
 * 32 batch rows x this many outputs on the fp32 matrix cores); the batched launches' tables count blocks in this unit. */
int sis_head_gemm_tile(void);

/* EqualLinear, model.py:152-162: out[b,o] = sum_i x[b,i] * w[o,i] * scale + bias[o]*lr_mul, then
 * (activation != 0) leaky-relu(0.2) * sqrt(2) as fused_leaky_relu does (fused_act.py:85-86).
 * x rows are x_row_stride floats apart (lets a [B, n_latent, D] latent be indexed in place). */
int sis_equal_linear(float* out, const float* x, int64_t x_row_stride, const float* w,
                     const float* bias, int batch, int in_dim, int out_dim, float scale,
                     float lr_mul, int activation, void* stream);

/* Every modulation EqualLinear of one Generator.forward in ONE launch (20 layers at 256^2; model.py:240 per
 * layer): out_base[row.out_offset + b*out_dim + o] = scale * <latent[b, row.latent_index, :], w[o, :]> + bias[o].
 * latent is [B, n_latent, dim]; table is a device array of n_layers rows of 8 int64:
 * {weight ptr, bias ptr, out offset (floats), latent index, out_dim, first_block, 0, 0}, first_block = running
 * sum of ceil(out_dim / sis_head_gemm_tile()); total_blocks = the final sum. */
int sis_modulation_batch(float* out_base, const float* latent, const int64_t* table, int n_layers,
                         int total_blocks, int batch, int n_latent, int dim, float scale, void* stream);

/* Every demodulation-coefficient row of one forward in ONE launch (see sis_modconv_demod); table rows of 8
 * int64: {wsq ptr, float bits of the conv scale, s offset, dscale offset, cout, first_block, cin, demodulate},
 * first_block = running sum of ceil(cout / sis_head_gemm_tile()). */
int sis_demod_batch(float* dscale_base, const float* s_base, const int64_t* table, int n_layers,
                    int total_blocks, int batch, void* stream);

/* Truncation trick, model.py:502-510: out = mean + psi * (w - mean); mean is [dim]. */
int sis_truncate(float* out, const float* w, const float* mean, float psi, int batch, int dim,
                 void* stream);

/* Weight prepack for the modulated convolution (once per checkpoint):
 *   wpk[ci][tap][co] = w[co][ci][tap]            (tap = kh*ks + kw)
 *   wsq[co][ci]      = sum_tap w[co][ci][tap]^2  (feeds the demodulation, model.py:243-245)
 * w is the ModulatedConv2d.weight parameter [1, Cout, Cin, ks, ks] (model.py:223-225). */
int sis_modconv_prepack(float* wpk, float* wsq, const float* w, int cout, int cin, int ksize,
                        void* stream);

/* dscale[b,co] = scale * rsqrt(scale^2 * sum_ci s[b,ci]^2 * wsq[co,ci] + 1e-8) when demodulate,
 * else scale  (model.py:241-245 with the style factored out of the weights; scale = 1/sqrt(Cin*ks^2),
 * model.py:219-220). */
int sis_modconv_demod(float* dscale, const float* s, const float* wsq, int batch, int cin,
                      int cout, float scale, int demodulate, void* stream);

/* Modulated convolution, stride 1, "same" padding, ks in {1,3}: model.py:272-276 (+ :287-292,
 * fused_act.py:85-86 when fuse_act != 0):
 *   y[b,co,h,w] = dscale[b,co] * sum_{ci,kh,kw} wpk[ci][kh*ks+kw][co] * s[b,ci] * x[b,ci,h+kh-p,w+kw-p]
 *   fuse_act:  y = lrelu_{0.2}(y + noise_weight[0]*noise[nb,0,h,w] + bias[co]) * sqrt(2)
 * noise is [1,1,H,W] (noise_batch_stride 0) or [B,1,H,W] (stride H*W) or NULL; noise_weight is a
 * DEVICE pointer to one float (the NoiseInjection.weight parameter).  MFMA fp32 (v_mfma_f32_32x32x2_f32).
 * workspace (optional, may be NULL): device scratch for split-K partial sums of launch-starved small
 * layers; without it those layers run unsplit.  Contents need not be preserved between calls. */
int sis_modconv2d(float* out, const float* x, const float* wpk, const float* s,
                  const float* dscale, const float* noise, int64_t noise_batch_stride,
                  const float* noise_weight, const float* bias, int batch, int cin, int cout,
                  int h, int w, int ksize, int fuse_act, const float* wino_u, void* workspace,
                  int64_t workspace_bytes, void* stream);

/* Winograd F(2x2,3x3) weight transform for sis_modconv2d's optional `wino_u` argument (ksize 3, even H and W,
 * Cin % 8 == 0): cin * 16 * cout floats holding (G w[co,ci] G^T)[xi], xi = 0..15, in the kernels' private operand order
 * [ci][q][ih][co][il][jj] with xi = 4 (2 ih + il) + 2 q + jj (opaque to the caller: only these kernels read it).  With
 * wino_u the stride-1 3x3 layers run 16 multiplies per 2x2 output tile instead of 36 (2.25x fewer MFMA FLOPs); NULL
 * selects the direct kernel. */
int sis_modconv_prepack_wino(float* u, const float* w, int cout, int cin, void* stream);

/* Modulated transposed convolution, stride 2, no padding, ks = 3: model.py:251-261 up to (not
 * including) the Blur: t[b,co,p,q] = dscale[b,co] * sum_{ci, 2h+kh=p, 2w+kw=q} wpk[ci][kh*3+kw][co]
 * * s[b,ci] * x[b,ci,h,w];  t is [B, Cout, 2H+1, t_row_stride] with the first 2W+1 floats of every row
 * meaningful (t_row_stride 0 = dense 2W+1; an even stride lets the kernel store output phase pairs as 8 bytes). */
int sis_modconv2d_up(float* t, const float* x, const float* wpk, const float* s,
                     const float* dscale, int batch, int cin, int cout, int h, int w,
                     int t_row_stride, void* workspace, int64_t workspace_bytes, void* stream);

/* The same transposed convolution in its fast-FIR form (csrc/modconv_upfir.hip): 25 instead of 36 multiplies per 2 x 2 input
 * positions (F(2,2) of the 2-tap even phases per axis), on v_mfma_f32_16x16x4_f32.  `u` = the 16 transformed weight planes,
 * [cin][8 plane pairs][cout][2], written by sis_modconv_up_fir_prepack from the layer's weight [cout][cin][3][3] (once per
 * checkpoint).
 * t: [batch][cout][2h+1][t_row_stride] with t_row_stride % 4 == 0 and >= 2w + 4 (columns beyond 2w are padding and receive
 * unspecified finite values).  sis_modconv_up_fir_supported: h, w >= 32, h even, w % 4 == 0, cin % 8 == 0, cout % 64 == 0,
 * operands below 2 GiB. */
int sis_modconv_up_fir_supported(int batch, int cin, int cout, int h, int w, int t_row_stride);
int sis_modconv_up_fir_prepack(float* u, const float* w, int cout, int cin, void* stream);
int sis_modconv2d_up_fir(float* t, const float* x, const float* u, const float* s, const float* dscale, int batch, int cin, int cout,
                         int h, int w, int t_row_stride, void* stream);

/* Blur (upfirdn2d up=1 down=1, model.py:89-92 / :262) fused with NoiseInjection + FusedLeakyReLU
 * (model.py:338-340): in [B,C,IH,IW] -> out [B,C,OH,OW], OH = IH + pad0 + pad1 - kh + 1.
 * fuse_act == 0 gives the plain blur. taps [kh,kw] float32 device pointer.  in_row_stride (floats, 0 = in_w)
 * lets the input rows be padded; sis_modconv2d_up writes rows of 2W+4 floats so that the 4x4 / pad-1 case
 * streams aligned 16-byte rows. */
int sis_blur_noise_act(float* out, const float* in, const float* taps, const float* noise,
                       int64_t noise_batch_stride, const float* noise_weight, const float* bias,
                       int batch, int channels, int in_h, int in_w, int in_row_stride, int kh, int kw,
                       int pad0, int pad1, int fuse_act, void* stream);

/* ToRGB, model.py:355-364: out[b,c,y,x] = scale * sum_ci w[c,ci]*s[b,ci]*x[b,ci,y,x] + bias[c]
 * + upfirdn2d(skip, taps, up=2, pad=(pad0,pad1))[b,c,y,x]   (skip == NULL: first ToRGB).
 * w is the [1,3,Cin,1,1] parameter, skip is [B,3,H/2,W/2], taps [kh,kw]. out channels fixed to
 * `cout` <= 4. */
int sis_to_rgb(float* out, const float* x, const float* w, const float* s, const float* bias,
               const float* skip, const float* taps, int batch, int cin, int cout, int h, int width,
               int kh, int kw, int pad0, int pad1, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Segmentation training step (float32): the non-convolution ends of EMANetUpdater.update_core /
 * TransUNetUpdater.update_core (updater/segmentation_updater.py:47-73, :83-106).
 */

/* EMANet loss tail, networks/ema_net/network.py:305-311 + CrossEntropyLoss2d :319-327:
 *   pred = bilinear(logits [B,C,h,w] -> [B,C,out_h,out_w], align_corners=True)
 *   loss[b] = mean_{y,x} NLL(log_softmax(pred)[b,:,y,x], labels[b,y,x]), ignore_index pixels count as 0.
 * workspace: sis_upsample_ce_workspace(batch, out_h, out_w) floats.  labels are int64 [B,out_h,out_w]. */
int sis_upsample_ce_workspace(int batch, int out_h, int out_w);
int sis_upsample_ce_fwd(float* loss, float* workspace, const float* logits, const int64_t* labels,
                        int batch, int classes, int h, int w, int out_h, int out_w,
                        int64_t ignore_index, void* stream);
/* grad_logits[b,c,i,j] = d(sum_b grad_loss[b] * loss[b]) / d logits[b,c,i,j] (deterministic gather). */
int sis_upsample_ce_bwd(float* grad_logits, const float* grad_loss, const float* logits,
                        const int64_t* labels, int batch, int classes, int h, int w, int out_h,
                        int out_w, int64_t ignore_index, void* stream);

/* torch.optim.SGD(momentum, dampening 0, no nesterov) for every tensor of an optimizer in one launch
 * (training_builder/ema_net_train_builder.py:27-48).  `table` is a device array of n_chunks rows of FIVE
 * int64: {param ptr, grad ptr, momentum-buffer ptr, count | (group << 48), bf16 shadow ptr or 0}, count <=
 * sis_sgd_chunk_elems().
 *   d = grad + wd[group]*p;  buf = first_step ? d : momentum*buf + d;  p -= lr[group]*buf;  shadow = bf16(p) if given
 * (the shadow is the bfloat16 copy of the fp32 master weight that bf16 layers read: no per-forward cast kernels)
 * lr / weight_decay are HOST arrays of n_groups (<= 4) floats. */
int sis_sgd_chunk_elems(void);
int sis_sgd_momentum(const int64_t* table, int n_chunks, const float* lr, const float* weight_decay,
                     int n_groups, float momentum, int first_step, void* stream);

/* The same step (not the first one: momentum buffers exist) with the hyper-parameters in DEVICE memory,
 * hyper = {lr[0..3], weight_decay[0..3], momentum}: the launch carries only pointers, so it can be captured in a
 * hipGraph and still follow a per-iteration LR schedule (the host rewrites `hyper` between replays). */
int sis_sgd_momentum_dev(const int64_t* table, int n_chunks, const float* hyper, void* stream);

/* emau.mu[i] = mu[i]*momentum + mean_b(mu_batch[b,i])*one_minus_momentum
 * (updater/segmentation_updater.py:56-66; in-place on the buffer, outside autograd). */
int sis_ema_update(float* mu, const float* mu_batch, float momentum, float one_minus_momentum,
                   int batch, int numel, void* stream);

/* A 3x3 convolution whose dilation (= padding) is half the image side (EMANet's dilation-16 bottleneck on 32 x 32 maps,
 * networks/ema_net/network.py:82-86) is one dense [4 Cin] -> [4 Cout] map per position of the half-size grid:
 * taps [4 cout][4 cin], rows (u, v, co), columns (u', v', ci), = weight[co][ci][u' - u + 1][v' - v + 1] (csrc/dilation_taps.hip);
 * the product itself runs on sis_conv1x1_f32*.  _bwd: dweight [cout][cin][3][3] from dtaps, fixed summation order. */
int sis_half_dilation_taps(float* taps, const float* weight, int cout, int cin, void* stream);
int sis_half_dilation_taps_bwd(float* dweight, const float* dtaps, int cout, int cin, void* stream);

/* EMANet's Expectation-Maximisation Attention Unit between its two 1x1 convolutions (networks/ema_net/network.py:219-249,
 * the no_grad block and the reconstruction):  mu <- mu0 (shared [channels][bases]) for every sample; `stages` rounds of
 * z = softmax_k(x^T mu), z_ = z / (1e-6 + sum_n z), mu = l2norm_c(x z_) (eps 1e-6);  x_out = relu(mu z^T), mu_out = mu.
 * x, x_out: [batch][channels][pixels] float32; mu_out: [batch][channels][bases].  fp32 MFMA products, 2 * stages + 1 launches,
 * no atomics (csrc/emau.hip).  bases must be 64, channels % 64 == 0, pixels % 128 == 0 (sis_emau_supported);
 * workspace: sis_emau_workspace_floats(...) floats (z, un-normalised bases, partial sums). */
int sis_emau_supported(int batch, int channels, int pixels, int bases);
int64_t sis_emau_workspace_floats(int batch, int channels, int pixels, int bases);
int sis_emau_forward(float* x_out, float* mu_out, const float* x, const float* mu0, float* workspace, int batch, int channels,
                     int pixels, int bases, int stages, void* stream);

/* Batch normalisation (training: batch statistics; F.batch_norm semantics incl. the unbiased running_var update)
 * fused with an optional residual add and ReLU, fp32 NCHW with H*W % 4 == 0
 * (networks/ema_net/network.py:37-56 bottleneck tail, :169-184 ConvBNReLU; bn_lib/nn/modules/batchnorm.py:51-56).
 * workspace: sis_bn_workspace_floats(batch, channels, hw) floats.  running_* may be NULL (no update).
 *   sis_bn_stats    mean[c], invstd[c] = 1/sqrt(var_biased + eps)   (+ running stats with `momentum`)
 *   sis_bn_act_fwd  y = [relu]( gamma*(x-mean)*invstd + beta [+ residual] )     (eval: pass running statistics)
 *   sis_bn_act_bwd  dy' = dy*[y>0] (when relu); dbeta = sum dy'; dgamma = sum dy'*xhat;
 *                   dx = gamma*invstd*(dy' - mean(dy') - xhat*mean(dy'*xhat)); dresidual = dy' (when not NULL).
 * relu_mask (may be NULL): sis_bn_mask_words(batch, channels, hw) 64-bit words, one sign bit per element of y, written by
 * sis_bn_act_fwd (relu != 0) and read by sis_bn_act_bwd INSTEAD of y (y may then be NULL): 1/8 byte per element for the ReLU
 * gate in both backward passes instead of 4. */
int64_t sis_bn_workspace_floats(int batch, int channels, int hw);
int64_t sis_bn_mask_words(int batch, int channels, int hw);
int sis_bn_stats(float* mean, float* invstd, float* running_mean, float* running_var, const float* x,
                 float* workspace, int batch, int channels, int hw, float eps, float momentum, void* stream);
int sis_bn_act_fwd(float* y, const float* x, const float* residual, const float* mean, const float* invstd,
                   const float* gamma, const float* beta, int batch, int channels, int hw, int relu, void* relu_mask,
                   void* stream);
int sis_bn_act_bwd(float* dx, float* dresidual, float* dgamma, float* dbeta, const float* dy, const float* y,
                   const float* x, const float* mean, const float* invstd, const float* gamma, float* workspace,
                   int batch, int channels, int hw, int relu, const void* relu_mask, void* stream);
/* Statistics + apply of the training-mode batch norm in ONE launch when a channel's batch * hw elements fit one workgroup's
 * registers (sis_bn_fused_supported: batch * hw <= 16 384 and hw % 256 == 0 -- every 32 x 32 layer of EMANet at batch 16 -- or, with
 * one workgroup of 1 024 threads per channel, batch * hw <= 65 536: the 64 x 64 layers; statistics then within 1e-6, not bitwise):
 * x is read once instead of twice; mean / invstd / running statistics as sis_bn_stats, y (and relu_mask) as sis_bn_act_fwd,
 * bitwise the same values.  sis_bn_act_bwd takes its single-pass form (dy and x read once; results to an ulp of the
 * three-launch form) under the same condition. */
int sis_bn_fused_supported(int batch, int channels, int hw);
int sis_bn_fused_fwd(float* y, float* mean, float* invstd, float* running_mean, float* running_var, const float* x,
                     const float* residual, const float* gamma, const float* beta, int batch, int channels, int hw, float eps,
                     float momentum, int relu, void* relu_mask, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dataset-loop neighbours of Generator.forward (SURVEY.md §8f "next" rows 1 and 2).
 */

/* Nearest k-means centre per pixel: labels[b,p] = argmin_k sum_c (x[b,c,p] - centres[k,c])^2, ties to the lowest
 * k (segmentation/gan_local_edit/factor_catalog.py:47-62 + :69-75: FactorCatalog.predict on [B,C,H,W]).
 * x [B,C,HW] float32, centres [K,C] float32 (K <= 64, 8 <= C < 8192), labels int64 [B,HW].
 * Bit-exact contract: the fp32 adds are associated exactly as torch's CPU ``.sum(dim=-1)`` associates them (the
 * reference's evaluation); the order is written out in csrc/dataset_ops.hip and oracle/kmeans_ref.py. */
int sis_kmeans_assign(int64_t* labels, const float* x, const float* centres, int batch, int channels,
                      int hw, int n_centres, void* stream);
/* The same with a workspace of sis_kmeans_workspace_ints(batch, hw) ints: the pixels the fast first pass leaves open are
 * listed there and the exact-order pass revisits exactly those, one per lane (without it: every 512-pixel workgroup that holds
 * one).  Same labels either way. */
int64_t sis_kmeans_workspace_ints(int batch, int hw);
int sis_kmeans_assign_ws(int64_t* labels, const float* x, const float* centres, int batch, int channels, int hw, int n_centres,
                         int* workspace, int64_t workspace_ints, void* stream);

/* float32 NCHW image in [-1,1] -> uint8 NHWC: clamp, (x+1)/2*255, truncating cast (the third-party make_image
 * called at create_dataset_for_segmentation.py:135; its rounding is not pinned by the reference). */
int sis_make_image_u8(uint8_t* out, const float* x, int batch, int channels, int hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * Plain 3x3 convolution (stride 1, padding 1, no bias) of the segmentation networks on the Winograd F(2x2,3x3)
 * MFMA kernel -- replaces F.conv2d / cuDNN for networks/ema_net/network.py's 3x3 layers (ConvBNReLU :169-184, the
 * deep stem :71-79, Bottleneck.conv2 :27-28 where stride = dilation = 1), forward and data gradient.
 * sis_conv3x3_prepack: w [Cout_w][Cin_w][3][3] -> u, the transformed weights the kernel streams.  adjoint = 0: the
 *   forward convolution (cout = Cout_w, cin = Cin_w).  adjoint = 1: the convolution that maps dL/dy to dL/dx
 *   (cout = Cin_w, cin = Cout_w, taps rotated by 180 degrees).  u holds cin * 16 * cout floats.
 * sis_conv3x3: out [B,cout,H,W] = conv(x [B,cin,H,W], u).  Needs W % 4 == 0, H % 2 == 0, cin % 8 == 0,
 *   cout % 4 == 0; workspace as for sis_modconv2d (split-K scratch, may be NULL). */
int sis_conv3x3_eligible(int batch, int cin, int cout, int h, int w); /* 1 if sis_conv3x3 takes this shape */
/* Weight gradient of the same convolution in the Winograd domain (csrc/conv_wgrad_wino.hip):
 *   dw [Cout][Cin][3][3] = d/dw of <gy, conv(x, w)>,  x [B,Cin,H,W], gy [B,Cout,H,W].
 * Needs even H and W, (B*H*W/4) % 8 == 0, Cin % 64 == 0, Cout % 64 == 0, tensors below 2^29 elements (32-bit byte offsets of
 * the masked buffer loads), fewer than 2^24 2x2 tiles and a workspace of >= 64 * Cin * Cout bytes (split-K slabs, added in
 * slice order: deterministic).  sis_conv3x3_wgrad_eligible returns 1 when the call would be taken. */
int sis_conv3x3_wgrad_eligible(int batch, int cin, int cout, int h, int w, int64_t workspace_bytes);
int sis_conv3x3_wgrad(float* dw, const float* x, const float* gy, int batch, int cin, int cout, int h, int w,
                      void* workspace, int64_t workspace_bytes, void* stream);
/* n_jobs layers of ONE shape (EMANet's repeated bottleneck units) through one tile launch + one finish launch: `dw`, `x`, `gy` are
 * HOST arrays of n_jobs device pointers; the K slices are planned for the layers' joint tile count. */
int sis_conv3x3_wgrad_multi(float* const* dw, const float* const* x, const float* const* gy, int n_jobs, int batch, int cin, int cout,
                            int h, int w, void* workspace, int64_t workspace_bytes, void* stream);
int sis_conv3x3_prepack(float* u, const float* w, int cout, int cin, int adjoint, void* stream);
/* forward and adjoint images of w [cout][cin][3][3] from one launch (training: the backward of the same step needs the adjoint):
 * u [cin][16][cout], u_adjoint [cout][16][cin]. */
int sis_conv3x3_prepack_both(float* u, float* u_adjoint, const float* w, int cout, int cin, void* stream);
/* Forward and adjoint images of SEVERAL layers from one launch.  `table`: device array of n_layers rows of 6 int64 --
 * (w pointer [cout][cin][3][3] float32, u pointer, u_adjoint pointer, cout, cin, first block of the layer) with
 * blocks per layer = ceil(cout * cin / 256); total_blocks = their sum. */
int sis_conv3x3_prepack_multi(const void* table, int n_layers, int total_blocks, void* stream);
int sis_conv3x3(float* out, const float* x, const float* u, int batch, int cin, int cout, int h, int w,
                void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight standardisation of StdConv2d (networks/trans_u_net/vit_seg_modeling_resnet_skip.py:20-27), one launch per
 * direction.  w [rows][n] float32 (rows = Cout, n = Cin*kh*kw).
 * fwd: w_hat (dtype out_dtype = SIS_F32 / SIS_F16 / SIS_BF16) = (w - mean_row) / sqrt(var_row + eps), biased variance;
 *      invstd [rows] = 1 / sqrt(var_row + eps).
 * bwd: dw [rows][n] float32 = invstd * (g - mean(g) - w_hat * mean(g * w_hat)), g = dL/dw_hat in grad_dtype. */
int sis_weight_std_fwd(void* w_hat, float* invstd, const float* w, int out_dtype, int rows, int n, float eps,
                       void* stream);
int sis_weight_std_bwd(float* dw, const void* grad_w_hat, const float* w, const float* invstd, int grad_dtype,
                       int rows, int n, float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * nn.GroupNorm (+ optional ReLU) of TransUNet's ResNetV2 trunk (vit_seg_modeling_resnet_skip.py:40-75,114-126),
 * x [B][C][hw] in x_dtype (SIS_F32 / SIS_F16 / SIS_BF16), fp32 arithmetic, affine gamma / beta [C] float32.
 * fwd: y (y_dtype = x_dtype or SIS_F32) = relu?((x - mean_g) * rstd_g * gamma_c + beta_c); mean / rstd [B*groups] kept.
 * residual (float32, may be NULL; needs y_dtype = SIS_F32): y = relu?(norm(x) + residual), the bottleneck's shortcut sum.
 * bwd: dx (x_dtype), dgamma / dbeta [C]; grad_y in g_dtype (= x_dtype or SIS_F32); the ReLU mask is recomputed from x, or
 *      taken from y_mask (the saved float32 output) when a residual was added; dresidual (float32, may be NULL) = masked grad;
 *      workspace (both directions): sis_group_norm_workspace_floats(B, C, hw) floats.
 *      y_lp (may be NULL; needs a float32 y of a 16-bit x): the same output also rounded to x's dtype in the same pass --
 *      the fp32 residual stream and the 16-bit tensor the next convolution reads; grad_y_lp (may be NULL): the gradient
 *      that came back through that copy, added to grad_y on load.
 *      counters (may be NULL): B*groups ints, ZERO on entry and left zero, private to the stream -- the per-group merge
 *      of the plane statistics then runs inside the statistics launch (the workgroup that completes a group does it, in
 *      channel order) and d(gamma) / d(beta) inside the apply launch: 2 launches per direction instead of 3 / 4; same
 *      values either way.
 *      relu_bits (may be NULL; fwd: with relu; sis_group_norm_gate_bytes bytes, 8-byte aligned): the forward leaves one bit per
 *      element ([y > 0], bit i of a flat bit array) and the backward gates on it instead of y_mask -- the residual form's
 *      backward then reads 1/8 byte instead of 4 per element and pass for the gate. */
int64_t sis_group_norm_workspace_floats(int batch, int channels, int hw);
int sis_group_norm_fwd(void* y, void* y_lp, float* mean, float* rstd, float* workspace, const void* x, const float* residual,
                       const float* gamma, const float* beta, int x_dtype, int y_dtype, int batch, int channels, int hw,
                       int groups, float eps, int relu, int* counters, void* relu_bits, void* stream);
int sis_group_norm_bwd(void* dx, float* dresidual, float* dgamma, float* dbeta, float* workspace, const void* grad_y,
                       const void* grad_y_lp, const void* x, const float* y_mask, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, int x_dtype, int g_dtype, int batch, int channels, int hw, int groups, int relu,
                       int* counters, const void* relu_bits, void* stream);
int64_t sis_group_norm_gate_bytes(int batch, int channels, int hw);
/* nn.BatchNorm2d in training mode (+ optional ReLU) with 16-bit or fp32 tensors -- the TransUNet decoder's Conv2dReLU
 * blocks (networks/trans_u_net/vit_seg_modeling.py:265-287).  Arguments as for group norm; statistics per channel over
 * (B, hw); mean / rstd [C]; running_mean / running_var (float32 [C], may both be NULL) are updated with `momentum`
 * (unbiased variance).  Workspace: sis_group_norm_workspace_floats(B, C, hw) floats.  `counters`: NULL, or C ints that are
 * zero before the call and zero again after it (private to the stream, as for group norm): the per-channel merge then
 * happens inside the statistics launch, by the workgroup that completes the channel, instead of in a launch of its own. */
int sis_batch_norm_fwd(void* y, float* mean, float* rstd, float* running_mean, float* running_var, float* workspace,
                       const void* x, const float* gamma, const float* beta, int x_dtype, int y_dtype, int batch,
                       int channels, int hw, float eps, float momentum, int relu, int* counters, void* stream);
int sis_batch_norm_bwd(void* dx, float* dgamma, float* dbeta, float* workspace, const void* grad_y, const void* x,
                       const float* mean, const float* rstd, const float* gamma, const float* beta, int x_dtype,
                       int g_dtype, int batch, int channels, int hw, int relu, int* counters, void* stream);

/* ------------------------------------------------------------------------------------------
 * Column sums out[c] = sum_r x[r][c] of a [rows][n] matrix (n % 4 == 0; x SIS_F32 / SIS_F16 / SIS_BF16, out float32): the
 * bias gradient of the encoder's Linear layers (networks/trans_u_net/vit_seg_modeling.py:53-110, what autograd computes as
 * grad_output.sum(0)); deterministic two-stage reduction.  workspace: sis_column_sum_workspace_floats(n) floats. */
int64_t sis_column_sum_workspace_floats(int n);
int sis_column_sum(float* out, float* workspace, const void* x, int x_dtype, int rows, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * nn.LayerNorm over the last dimension (TransUNet encoder, networks/trans_u_net/vit_seg_modeling.py:171-190,233-250),
 * x [rows][n], n in {256, 512, 768, 1024}, x / y / grad dtypes SIS_F32 or SIS_BF16 (fp32 arithmetic), gamma / beta [n].
 * fwd: y = (x - mean_row) * rstd_row * gamma + beta; mean / rstd [rows] kept.
 * bwd: dx (x's dtype), dgamma / dbeta [n]; workspace: sis_layer_norm_workspace_floats(n) floats. */
int sis_layer_norm_workspace_floats(int n);
int sis_layer_norm_fwd(void* y, float* mean, float* rstd, const void* x, const float* gamma, const float* beta, int x_dtype,
                       int y_dtype, int rows, int n, float eps, void* stream);
int sis_layer_norm_bwd(void* dx, float* dgamma, float* dbeta, float* workspace, const void* grad_y, const void* x,
                       const float* mean, const float* rstd, const float* gamma, int x_dtype, int g_dtype, int rows, int n,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * nn.UpsamplingBilinear2d (align_corners=True) of the TransUNet decoder
 * (networks/trans_u_net/vit_seg_modeling.py:290-329), forward and backward, f32 / f16 / bf16 (dtype = SIS_*).
 * backward = 0: out [planes][out_h][out_w] from x [planes][h][w].
 * backward = 1: out = grad_x [planes][h][w] from x = grad_out [planes][out_h][out_w] (gather, deterministic). */
int sis_upsample_bilinear(void* out, const void* x, int dtype, int64_t planes, int h, int w, int out_h, int out_w,
                          int backward, void* stream);
/* The same with an image stride on the UPSAMPLED-size tensor (elements between consecutive images; >= channels*out_h*out_w):
 * forward writes its result into the leading channels of a wider tensor, backward reads grad_out from there -- the decoder's
 * torch.cat([up(x), skip], dim=1) (vit_seg_modeling.py:300-303) then costs one copy of the skip feature only.  Strided
 * tensors need the x2 case (out_h = 2h, out_w = 2w). */
int sis_upsample_bilinear_strided(void* out, const void* x, int dtype, int batch, int channels, int h, int w, int out_h, int out_w,
                                  int64_t image_stride, int backward, void* stream);

/* Weight gradient of the same 1x1 convolutions (networks/hip_conv.py::_Pointwise.backward; reference layers as for
 * sis_conv1x1_f32): dw[co][ci] = sum_{b, p} gy[b][co][p] * x[b][ci][p], fp32 on the matrix cores, NCHW rows straight from
 * HBM by LDS-DMA, split-K over the (sample, pixel) stream with an ordered (deterministic) reduction through `workspace`
 * (sis_conv1x1_wgrad_f32_workspace bytes; fewer bytes = fewer slices, NULL = one).  Supported: hw % 64 == 0 and
 * (cout % 128 == 0 and cin % 64 == 0) or (cout % 64 == 0 and cin % 128 == 0) or (cout % 256 == 0 and cin % 32 == 0). */
int sis_conv1x1_wgrad_f32_supported(int batch, int cin, int cout, int hw);
int64_t sis_conv1x1_wgrad_f32_workspace(int batch, int cin, int cout, int hw);
int sis_conv1x1_wgrad_f32(float* dw, const float* gy, const float* x, int batch, int cin, int cout, int hw,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* (as sis_conv3x3_wgrad_multi; workspace: any size, the slices adapt -- sis_conv1x1_wgrad_f32_workspace x n_jobs is plenty) */
int sis_conv1x1_wgrad_f32_multi(float* const* dw, const float* const* gy, const float* const* x, int n_jobs, int batch, int cin, int cout,
                                int hw, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * nn.MaxPool2d / F.max_pool2d of the segmentation backbones (ceil_mode false, dilation 1): EMANet stem (3, 2, 1)
 * (networks/ema_net/network.py:66), TransUNet root (3, 2, 0) (vit_seg_modeling_resnet_skip.py:146); f32 / f16 / bf16.
 * backward = 0: out [planes][out_h][out_w] and argmax (one byte per output: kh * kernel + kw of the FIRST maximum in
 *               row-major window order, ATen's tie rule) from x [planes][h][w].
 * backward = 1: out = grad_x [planes][h][w] from x = grad_out [planes][out_h][out_w] and the forward's argmax (gather,
 *               deterministic, bit-equal to ATen's max_pool2d_with_indices_backward). */
int sis_max_pool2d(void* out, unsigned char* argmax, const void* x, int dtype, int64_t planes, int h, int w, int out_h,
                   int out_w, int kernel, int stride, int padding, int backward, void* stream);

/* ------------------------------------------------------------------------------------------
 * Patch-wise page inference.  The patch grid is the product of `nx` left edges `xs` and
 * `ny` top edges `ys` (device int32 arrays, ascending), patch n = yi * nx + xi, as
 * segmentation/analysis_segmenter.py:83-113 enumerates them.
 * sis_crop_patches_u8: image uint8 [height][width][channels] -> out float32 [ny*nx][channels][patch][patch] =
 *   ((u8 / 255) - 0.5) / 0.5 with zeros read outside the image (PIL crop + ToTensor + Normalize, :115-130).
 * sis_assemble_max: pred float32 [ny*nx][classes][patch][patch] -> out [classes][height][width] = maximum over
 *   the covering patches (:147-167); labels (uint8 [height][width], may be NULL) = first maximal class
 *   (networks/base_segmenter.py:59-62).  classes / channels <= 16. */
int sis_crop_patches_u8(float* out, const unsigned char* image, const int* xs, const int* ys, int nx, int ny,
                        int height, int width, int channels, int patch, void* stream);
int sis_assemble_max(float* out, unsigned char* labels, const float* pred, const int* xs, const int* ys, int nx,
                     int ny, int classes, int height, int width, int patch, void* stream);

/* ------------------------------------------------------------------------------------------
 * bf16 convolutions of the segmentation training step on the matrix cores, direct on NCHW tensors (csrc/conv_bf16.hip):
 * TransUNet under bf16 autocast -- StdConv2d 3x3 / 1x1 (networks/trans_u_net/vit_seg_modeling_resnet_skip.py:20-37,40-75),
 * decoder Conv2dReLU and segmentation head (vit_seg_modeling.py:265-287,324-329), patch embedding (:125-168).
 * Kernel sizes 1 and 3 (padding ksize/2), strides 1 and 2, Cin % 16 == 0 (% 64 for 1x1 stride 1); x / y bf16 NCHW,
 * fp32 accumulation, optional fp32 bias[cout] added before the rounding to bf16.
 *   sis_conv_bf16_supported     1 when a tile plan exists for the layer.
 *   sis_conv_bf16_packed_elems  bf16 elements of the packed weight image (-1: unsupported).
 *   sis_conv_bf16_pack          weight [cout][cin][k][k] (SIS_F32 or SIS_BF16) -> the kernel's LDS image, per output
 *                               channel tile / input channel chunk / tap / row, 16-byte units XOR-swizzled.
 *                               adjoint = 1 packs the convolution that takes dL/dy to dL/dx of a stride-1 layer
 *                               (channel roles swapped, taps rotated by 180 degrees); it is then used with
 *                               sis_conv_bf16(cin = the layer's cout, cout = the layer's cin).
 *   sis_conv_bf16               y [batch][cout][ho][wo] = conv(x [batch][cin][h][w], packed) (+ bias). */
int sis_conv_bf16_supported(int cin, int cout, int h, int w, int ksize, int stride);
int64_t sis_conv_bf16_packed_elems(int cin, int cout, int h, int w, int ksize, int stride, int adjoint);
int sis_conv_bf16_pack(void* packed, const void* weight, int weight_dtype, int cin, int cout, int h, int w, int ksize,
                       int stride, int adjoint, void* stream);
int sis_conv_bf16(void* y, const void* x, const void* packed, const float* bias, int batch, int cin, int cout, int h, int w,
                  int ksize, int stride, void* stream);
/* All StdConv2d layers of a network in ONE launch: standardise every filter (as sis_weight_std_fwd, bf16 result) and write it
 * into w_hat, into the forward image and (stride-1 layers) into the adjoint image of its layer
 * (networks/trans_u_net/vit_seg_modeling_resnet_skip.py:20-27: the reference standardises every weight in every forward).
 * `table`: device array of 14 int64 per layer -- pointers w (float32 [cout][cin][k][k]), w_hat (bf16), invstd (float32 [cout]),
 * packed, adjoint (0 = none), then cout, cin, k, mt, kc, mt2 (from sis_weight_std_pack_plan), rows = ceil(cout / mt) * mt,
 * row_begin = sum of the previous layers' rows, filter_begin = sum of the previous layers' cout; total_rows = sum of rows.  sis_weight_std_pack_plan returns 0 for a layer the
 * bf16 convolution kernels have no plan for. */
int sis_weight_std_pack_plan(int cin, int cout, int ksize, int stride, int* mt, int* kc, int* mt2, int64_t* packed_elems,
                             int64_t* adjoint_elems);
int sis_weight_std_pack_multi(const void* table, int n_layers, int total_rows, float eps, void* stream);
/* ... and its backward for all layers in one launch (per 64 layers): grads[i] = dL/dw_hat of layer i (bf16, device pointer, NULL =
 * no gradient for that layer), dw[i] = float32 result of the layer's weight shape; `grads`, `dw`, `couts` are HOST arrays of
 * n_layers entries in the table's layer order. */
int sis_weight_std_bwd_multi(const void* table, const void* const* grads, void* const* dw, const int* couts, int n_layers,
                             void* stream);
/* forward and adjoint packing of a stride-1 layer's weight in one launch (training: both are needed every step) */
int sis_conv_bf16_pack_both(void* packed, void* packed_adjoint, const void* weight, int weight_dtype, int cin, int cout, int h,
                            int w, int ksize, void* stream);

/* Weight gradient of a stride-1, padding-1 3x3 bf16 convolution (csrc/conv_bf16_wgrad.hip):
 * dw [cout][cin][3][3] (SIS_F32 or SIS_BF16) = sum_{n,y,x} grad_y[n][co][y][x] * x[n][ci][y+ky-1][x+kx-1], x / grad_y bf16
 * NCHW, fp32 accumulation; split-K partial tiles go through `workspace` and are added in a fixed order (deterministic).
 * sis_conv_bf16_wgrad_supported: 1 when a tile plan exists (cin, cout >= 32) and its slabs fit `workspace_bytes`. */
int sis_conv_bf16_wgrad_supported(int batch, int cin, int cout, int h, int w, int64_t workspace_bytes);
int sis_conv_bf16_wgrad(void* dw, int dw_dtype, const void* x, const void* grad_y, int batch, int cin, int cout, int h, int w,
                        void* workspace, int64_t workspace_bytes, void* stream);
/* The 1x1 stride-1 layers (conv1x1 / StdConv2d 1x1 of every bottleneck, vit_seg_modeling_resnet_skip.py:30-37,40-75; the
 * reference's autograd runs convolution_backward): dw [cout][cin] (SIS_F32 or SIS_BF16) = sum_{n,p} grad_y[n][co][p] * x[n][ci][p],
 * x [batch][cin][pixels] / grad_y [batch][cout][pixels] bf16 (NCHW with the plane flattened; any plane size, odd ones take a
 * funnel-shift load path), fp32 accumulation, unit slabs through `workspace` added in unit order (deterministic). */
int sis_conv1x1_bf16_wgrad_supported(int batch, int cin, int cout, int pixels, int64_t workspace_bytes);
int sis_conv1x1_bf16_wgrad(void* dw, int dw_dtype, const void* x, const void* grad_y, int batch, int cin, int cout, int pixels,
                           void* workspace, int64_t workspace_bytes, void* stream);
/* Both weight-gradient kernels for n_jobs layers of ONE shape (the trunk's repeated bottleneck units, queued during the
 * backward): `dw`, `x`, `grad_y` are HOST arrays of n_jobs device pointers; one tile launch (grid.z = layer) and one reduction
 * launch per <= 16 layers.  The tile plan counts the layers' workgroups together: each layer is cut into fewer, longer units than
 * it would be alone (less slab traffic, fewer prologues).  Results are those of the single-layer calls up to the order of the
 * fp32 partial sums. */
int sis_conv_bf16_wgrad_multi(void* const* dw, int dw_dtype, const void* const* x, const void* const* grad_y, int n_jobs, int batch,
                              int cin, int cout, int h, int w, void* workspace, int64_t workspace_bytes, void* stream);
int sis_conv1x1_bf16_wgrad_multi(void* const* dw, int dw_dtype, const void* const* x, const void* const* grad_y, int n_jobs, int batch,
                                 int cin, int cout, int pixels, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * The root convolution of the hybrid trunk (networks/trans_u_net/vit_seg_modeling_resnet_skip.py:115-125: StdConv2d(3, 64,
 * kernel_size=7, stride=2, padding=3) on the image) on the bf16 matrix cores: forward and weight gradient (the image needs no
 * gradient).  x [B][3][h][w] float32 or bfloat16 (converted while staged), y / grad_y [B][64][ho][wo] bfloat16 with
 * ho = (h - 1) / 2 + 1.  `packed`: sis_stem_conv_packed_elems() bf16 elements written by sis_stem_conv_pack from the
 * (standardised) weight [64][3][7][7] (float32 or bfloat16).  dw [64][3][7][7] float32 or bfloat16; the partial sums of the
 * workgroups go through `workspace` (sis_stem_conv_wgrad_workspace_bytes) and are added in a fixed order. */
int sis_stem_conv_supported(int cin, int cout, int ksize, int stride, int padding, int h, int w);
int64_t sis_stem_conv_packed_elems(void);
int sis_stem_conv_pack(void* packed, const void* weight, int weight_dtype, void* stream);
int sis_stem_conv_fwd(void* y, const void* x, int x_dtype, const void* packed, int batch, int h, int w, void* stream);
int64_t sis_stem_conv_wgrad_workspace_bytes(int batch, int h, int w);
int sis_stem_conv_wgrad(void* dw, int dw_dtype, const void* x, int x_dtype, const void* grad_y, int batch, int h, int w, void* workspace,
                        int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * fp32 pointwise (1x1, stride 1) convolution of the EMANet training step (csrc/conv1x1_f32.hip;
 * networks/ema_net/network.py:24,29,106-107 Bottleneck conv1 / conv3 / downsample, :219-249 EMAU, :271-289 fc0 / fc1),
 * exact fp32 on v_mfma_f32_32x32x2_f32, NCHW in and out.
 *   data_gradient = 0:  y [batch][cout][hw] = weight [cout][cin] . x [batch][cin][hw] (+ bias[cout])
 *   data_gradient = 1:  y = dL/dx [batch][cin][hw] = weight^T . x, x = dL/dy [batch][cout][hw] (the weight tensor itself is
 *                       read as the k-major operand: no transposed copy)
 * Contraction length % 32 == 0, output channels % 4 == 0, hw % 4 == 0, 16-byte aligned pointers. */
int sis_conv1x1_f32_supported(int cin, int cout, int hw);
int sis_conv1x1_f32(float* y, const float* x, const float* weight, const float* bias, int batch, int cin, int cout, int hw,
                    int data_gradient, void* stream);
/* dx [batch][cin][hw] = weight^T . dy + skip_grad: the data gradient of a bottleneck's first 1x1 convolution with the gradient
 * that arrives over the identity shortcut (networks/ema_net/network.py:37-56: out = relu(bn3(...) + x)) added in the epilogue
 * instead of by a separate element-wise pass. */
int sis_conv1x1_f32_dgrad_add(float* dx, const float* dy, const float* weight, const float* skip_grad, int batch, int cin, int cout,
                              int hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * bf16 GEMM with the Linear layers' element-wise tail fused into the epilogue (csrc/gemm_bf16.hip): the ViT encoder of
 * TransUNet under bf16 autocast -- Attention.query/key/value/out (networks/trans_u_net/vit_seg_modeling.py:60-67,76-96),
 * Mlp fc1 / GELU / dropout / fc2 / dropout (:104-122), the Block residual adds (:181-189); replaces F.linear (a library
 * GEMM) + separate bias / GELU / dropout / add / cast kernels, forward and backward.
 *   C[m][n] = epilogue( sum_k opA[m][k] * opB[k][n] ),  bf16 operands, fp32 accumulation, row-major C with leading dim ldc
 *   layout 0 (NT): A [m][k] (lda), B [n][k] (ldb)        forward:          y  = x W^T
 *   layout 1 (NN): A [m][k] (lda), B [k][n] (ldb)        data gradient:    dx = g W
 *   layout 2 (TN): A [k][m] (lda), B [k][n] (ldb)        weight gradient:  dW = g^T x   (k = tokens; any k >= 1)
 * epilogue:
 *   SIS_GEMM_EPI_NONE             C bf16 = acc
 *   SIS_GEMM_EPI_BIAS             C bf16 = acc + bias[n]
 *   SIS_GEMM_EPI_BIAS_GELU_DROP   with h = acc + bias:  C bf16 = dropout(gelu(h))  (erf GELU);  C2 bf16 = d C / d h = gelu'(h) * the same
 *                                 dropout factor -- what SIS_GEMM_EPI_GELU_BWD of the backward multiplies by (round 5: it was h, and the
 *                                 backward recomputed erf, exp and the dropout hash per element)
 *   SIS_GEMM_EPI_BIAS_DROP_RESID  C fp32 = resid[m][n] + dropout(acc + bias)      (resid fp32, leading dim ldc)
 *   SIS_GEMM_EPI_GELU_BWD         C bf16 = acc * pre[m][n]   (pre bf16, leading dim ldc: the C2 of the forward's SIS_GEMM_EPI_BIAS_GELU_DROP;
 *                                 seed / site / drop_p are not read)
 *   SIS_GEMM_EPI_F32              C fp32 = acc; with splits > 1 the K range is cut into `splits` slices whose partial results
 *                                 go through `workspace` and are added in slice order (deterministic)
 * dropout: element (m, n) of site `site` is dropped iff hash(seed word, site, m * n_cols + n) < drop_p * 2^32, else scaled by
 * 1 / (1 - drop_p); the forward and the backward launch of a site evaluate the same function (no stored mask).  `seed`
 * points to a 64-bit device word (sis_dropout_advance steps it once per iteration).  drop_p = 0: no dropout.
 * bias_seg > 0: the bias is three vectors of bias_seg = n / 3 entries (bias, bias1, bias2: the query | key | value biases of the
 * fused projection stay three parameters); bias_seg = 0: one vector `bias`.
 * tile (BM x BN x BK, LDS stages): 0 = 128x128x64 / 2, 1 = 256x128x64 / 2, 2 = 128x256x64 / 2, 3 = 256x256x64 / 2,
 * 4 = 128x128x32 / 3, 5 = 128x128x32 / 4, 6 = 128x128x64 / 3, 7 = 256x128x64 / 3 (one wave per 64x64 sub-tile),
 * 8 = 128x96x64 / 2 (NT only; waves of 64x48: n = 768 / 2304 give 512 / 1536 tiles = whole rounds of 2 workgroups per CU).  NT / NN need k % 64 == 0; n, ldc % 4 == 0;
 * lda, ldb % 8 == 0; 16-byte aligned pointers.  m (and k for TN) need not be tile multiples. */
#define SIS_GEMM_EPI_NONE 0
#define SIS_GEMM_EPI_BIAS 1
#define SIS_GEMM_EPI_BIAS_GELU_DROP 2
#define SIS_GEMM_EPI_BIAS_DROP_RESID 3
#define SIS_GEMM_EPI_GELU_BWD 4
#define SIS_GEMM_EPI_F32 5
/* tile codes 9..11: 256 x 96 / 256 x 192 / 256 x 288 output tiles, one 8-wave workgroup per compute unit, register-double-
 * buffered fragments over four LDS stages (csrc/gemm256_bf16.hip): layout NT only, splits == 1, k % 64 == 0, k >= 128, every
 * epilogue except SIS_GEMM_EPI_F32.  8 192-token GEMMs of width 768 / 3072 / 2304 give 256 / 512 / 256 such tiles. */
#define SIS_GEMM_TILE_256X96 9
#define SIS_GEMM_TILE_256X192 10
#define SIS_GEMM_TILE_256X288 11
int64_t sis_gemm_bf16_workspace_bytes(int m, int n, int splits);
int sis_gemm_bf16(void* c, void* c2, const void* a, const void* b, int layout, int epilogue, int m, int n, int k, int lda,
                  int ldb, int ldc, const float* bias, const float* bias1, const float* bias2, int bias_seg, const float* resid,
                  const void* pre, const void* seed, int site, float drop_p, int splits, void* workspace,
                  int64_t workspace_bytes, int tile, void* stream);
/* Weight gradient AND bias gradient of a Linear layer (what autograd computes as grad.t() @ x and grad.sum(0)) in the two
 * launches of the split-K weight gradient: dw [m][n] float32 (TN layout: grad [k][m], x [k][n], splits > 1), db [m] float32 =
 * column sums of grad by extra workgroups of the GEMM launch (dispatched last: they fill the slots its final, partly filled round
 * of tiles leaves idle) and of the slab reduction; sis_column_sum's summation order.
 * workspace: sis_gemm_bf16_workspace_bytes(m, n, splits) + 64 * m * 4 bytes; tile 0 or 4..6. */
int sis_gemm_bf16_wgrad_bias(void* dw, float* db, const void* grad, const void* x, int m, int n, int k, int lda, int ldb, int splits,
                             void* workspace, int64_t workspace_bytes, int tile, void* stream);
/* The same for n_jobs Linear layers of ONE shape (the encoder's twelve blocks, queued during the backward): one launch for the
 * n_jobs products -- every problem contracts its whole K per tile: no split-K, no slabs, no reduction launch -- and one for the
 * column sums.  `dw`, `db`, `grad`, `x`: HOST arrays of n_jobs device pointers (dw[j] [m][n] float32, db[j] [m] float32, grad[j]
 * bf16 [k][lda], x[j] bf16 [k][ldb]).  tile: a 128 x 128 four-wave tile code (0, 4, 5, 6).  workspace:
 * sis_gemm_bf16_wgrad_multi_workspace_bytes(n_jobs, m, k) bytes. */
int64_t sis_gemm_bf16_wgrad_multi_workspace_bytes(int n_jobs, int m, int k);
int sis_gemm_bf16_wgrad_bias_multi(void* const* dw, float* const* db, const void* const* grad, const void* const* x, int n_jobs, int m,
                                   int n, int k, int lda, int ldb, void* workspace, int64_t workspace_bytes, int tile, void* stream);

/* Transposed bf16 copies of `n_tensors` matrices by one launch (csrc/vit_elementwise.hip): dst_i [cols_i][rows_i] =
 * src_i [rows_i][cols_i]^T.  `table`: DEVICE array of n_tensors rows of 5 int64 {src, dst, rows, cols, first_tile} with
 * first_tile = running sum of ceil(rows / 64) * ceil(cols / 64); total_tiles = the final sum.  Used for the transposed
 * weight shadows that turn a Linear layer's data gradient into an NT product (sis_gemm_bf16 tiles 9..11). */
int sis_transpose_bf16_multi(const void* table, int n_tensors, int total_tiles, void* stream);
/* dst[b][c][r] = src[b][r][c] for 2- or 4-byte elements: `x.flatten(2).transpose(-1, -2)` of the patch embeddings
 * (networks/trans_u_net/vit_seg_modeling.py:151-153), `hidden_states.permute(0, 2, 1).contiguous()` of DecoderCup.forward
 * (:341-344), and the gradients of both (the same operation). */
int sis_transpose_batched(void* dst, const void* src, int elem_bytes, int batch, int rows, int cols, void* stream);

/* The same kernel over `batches` problems (grid.y): A / B / C of entry i start i * {a,b,c}_batch_stride elements after the base
 * pointers (stride 0 = shared operand).  With sum_over_batches != 0 the entries are the K slices of ONE result instead
 * (C fp32 [m][n] = sum_i op(A_i) op(B_i), ordered slab reduction through `workspace`; batches must be 1, 2, 4 or a multiple
 * of 8).  This is how the bf16 1x1 convolutions of TransUNet's ResNetV2 trunk run on NCHW tensors
 * (networks/trans_u_net/vit_seg_modeling_resnet_skip.py:30-37 conv1x1 / StdConv2d): per image
 *   forward          y_i [cout][hw] = W [cout][cin] . x_i [cin][hw]        layout NN, A shared
 *   data gradient    dx_i [cin][hw] = W^T . dy_i                           layout TN, A = W read K-major, shared
 *   weight gradient  dW [cout][cin] = sum_i dy_i [cout][hw] . x_i^T        layout NT, summed over the images
 * epilogue: SIS_GEMM_EPI_NONE (bf16) or SIS_GEMM_EPI_F32. */
int sis_gemm_bf16_batched(void* c, const void* a, const void* b, int layout, int epilogue, int m, int n, int k, int lda, int ldb,
                          int ldc, int batches, int64_t a_batch_stride, int64_t b_batch_stride, int64_t c_batch_stride,
                          int sum_over_batches, void* workspace, int64_t workspace_bytes, int tile, void* stream);

/* Dropout stream of the fused ViT-encoder kernels (csrc/vit_elementwise.hip; nn.Dropout of vit_seg_modeling.py:70-71,108,138).
 *   sis_dropout_advance   steps the 64-bit device seed word once per training iteration (graph-capturable).
 *   sis_dropout_bwd_cast  out bf16[numel] = grad fp32[numel] * dropout_factor(seed, site, element index): the gradient of
 *                         `resid + dropout(y)` w.r.t. y for a site whose forward ran as SIS_GEMM_EPI_BIAS_DROP_RESID with the
 *                         same site id and a dense [m][n] output (element index = m * n_cols + n). */
int sis_dropout_advance(void* seed, void* stream);
int sis_dropout_bwd_cast(void* out, const float* grad, int64_t numel, const void* seed, int site, float drop_p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused multi-head self-attention of the ViT encoder, head size 64 (csrc/attention_bf16.hip).
 * Replaces Attention.forward's  softmax(q k^T / sqrt(d)) v  with its transposes / permutes
 * (networks/trans_u_net/vit_seg_modeling.py:71-74,83-94; attention dropout rate 0.0, vit_seg_configs.py:16) and its backward.
 *   qkv    bf16 [batch][n][3 * heads * 64]   the fused query | key | value projection, read in place (head h = columns h*64..)
 *   ctx    bf16 [batch][n][heads * 64]       context, heads merged (what Attention.out consumes)
 *   lse    fp32 [batch][heads][n]            log-sum-exp of the scaled scores (saved for the backward)
 *   d_qkv  bf16, layout of qkv; d_ctx bf16, layout of ctx; delta fp32 [batch][heads][n] scratch (rowsum(d_ctx o ctx))
 * No [n][n] tensor is written; the backward accumulates nothing across workgroups (no atomics: bitwise reproducible). */
/* LayerNorm backward (csrc/layer_norm.hip) with the residual-block fusions of the ViT encoder (vit_seg_modeling.py:181-189):
 * dx = residual_grad + LN'(grad_y) (residual_grad fp32 or NULL), and optionally cast_out bf16 = dx * dropout factor of the
 * dropout site the previous residual add used (what sis_dropout_bwd_cast computes, without its pass over memory). */
int sis_layer_norm_bwd_fused(void* dx, float* dgamma, float* dbeta, float* workspace, const void* grad_y, const void* x,
                             const float* mean, const float* rstd, const float* gamma, int x_dtype, int g_dtype, int rows, int n,
                             const float* residual_grad, void* cast_out, const void* seed, int site, float drop_p, void* stream);
/* Deferred form of the LayerNorm parameter gradients (a ViT encoder has 25 norms; nothing reads d(gamma) / d(beta) before the
 * optimizer or the gradient exchange): sis_layer_norm_bwd_fused_partial = sis_layer_norm_bwd_fused without its reduction launch
 * (the sis_layer_norm_bwd_parts(rows) partial rows stay in `workspace`), sis_layer_norm_param_reduce_multi = the reductions of
 * `count` such jobs in one launch per 32 jobs (HOST arrays of device pointers / ints), bitwise the results of the undeferred call. */
int sis_layer_norm_bwd_fused_partial(void* dx, float* workspace, const void* grad_y, const void* x, const float* mean,
                                     const float* rstd, const float* gamma, int x_dtype, int g_dtype, int rows, int n,
                                     const float* residual_grad, void* cast_out, const void* seed, int site, float drop_p,
                                     void* stream);
int sis_layer_norm_bwd_parts(int rows);
int sis_layer_norm_param_reduce_multi(void* const* dgamma, void* const* dbeta, const void* const* part, const int* n_part,
                                      const int* n, int count, void* stream);
int sis_attention_fwd(void* ctx, float* lse, const void* qkv, int batch, int n, int heads, void* stream);
int sis_attention_bwd(void* d_qkv, float* delta, const void* d_ctx, const void* qkv, const void* ctx, const float* lse, int batch,
                      int n, int heads, void* stream);

/* ------------------------------------------------------------------------------------------
 * TransUNet objective, fused (csrc/loss_ops.hip):  0.5 * CrossEntropy + 0.5 * Dice(softmax)  of
 * updater/segmentation_updater.py:95-102 with networks/trans_u_net/utils.py:7-42 (DiceLoss, smooth 1e-5, sums over the batch).
 *   logits [batch][classes][hw] (SIS_F32 or SIS_BF16), labels int64 [batch][hw]; 2 <= classes <= 8, hw % 4 == 0.
 *   fwd: out3 = {combined, cross entropy, dice}; stats (1 + 2 * classes floats) = constants for the backward;
 *        workspace: sis_ce_dice_workspace_floats(classes) floats (per-workgroup partial sums, added in order: deterministic).
 *   bwd: grad_logits (dtype of logits) = d combined / d logits * grad_loss[0] (grad_loss NULL = 1).
 * Labels outside [0, classes) are ignored by the cross entropy and are an all-zero one-hot row for the Dice sums. */
int sis_ce_dice_workspace_floats(int classes);
int sis_ce_dice_fwd(float* out3, float* stats, float* workspace, const void* logits, int dtype, const int64_t* labels, int batch,
                    int classes, int hw, void* stream);
int sis_ce_dice_bwd(void* grad_logits, const void* logits, int dtype, const int64_t* labels, const float* stats,
                    const float* grad_loss, int batch, int classes, int hw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIS_HIP_H */
