/*
 * wcmc_hip.h -- C ABI of libwcmc_hip.so, the MI355X (gfx950) hot path of the
 * KPCN-Manifold training step (Mephisto405/WCMC).
 *
 * The reference is pure Python on PyTorch; the arithmetic of its hot path lives
 * in torch.nn / cuDNN and in the external `sbmc` package's Halide ops.  There is
 * no FFI in the reference for this path, so each entry point below names the
 * reference EXPRESSION it replaces (file:line under the reference tree) and is
 * what a ctypes binding under support/ would bind.
 *
 * Conventions
 *   - All tensors are fp32 device memory owned by the caller (PyTorch).  The
 *     library never allocates, frees or retains a pointer past return.
 *   - "NHWC view" = (ptr, N, H, W, C, sn, sh, sw): element (n,y,x,c) lives at
 *     ptr[n*sn + y*sh + x*sw + c]; the channel stride is 1.  ptr must be 16-byte
 *     aligned, sn/sh/sw multiples of 4, and sw >= round_up(C, 4) so that a 16-byte
 *     access that starts at a channel multiple of 4 stays inside the pixel.
 *     Slices of wider buffers (concat targets, cropped images) are expressed by
 *     the strides.
 *   - Every call only enqueues work on `stream` (a hipStream_t) and returns; no
 *     host synchronisation, no global mutable state except a thread-local error
 *     string.
 *   - Return value: 0 on success, a negative wcmc_status on failure;
 *     wcmc_last_error() then describes it.  Nothing aborts or throws.
 */
#ifndef WCMC_HIP_H
#define WCMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WCMC_ABI_VERSION 2     /* 2: `terms` of the GEMM entry points, packing mode 2 (round 3); round 4 widened the value ranges only (terms = 1, mode 3) */

enum wcmc_status {
  WCMC_OK = 0,
  WCMC_ERR_BAD_ARG = -1,     /* null pointer, negative size, unsupported shape */
  WCMC_ERR_ALIGNMENT = -2,   /* pointer or stride violates the NHWC-view contract */
  WCMC_ERR_WORKSPACE = -3,   /* workspace too small */
  WCMC_ERR_LAUNCH = -4       /* hipGetLastError() after a launch */
};

enum wcmc_act { WCMC_ACT_LINEAR = 0, WCMC_ACT_RELU = 1, WCMC_ACT_LEAKY_RELU = 2 };

int wcmc_abi_version(void);
const char* wcmc_last_error(void);

/* ---------------------------------------------------------------- layout helpers
 * Strided copies between the reference's NCHW tensors (batch dict entries,
 * support/datasets.py:1080-1126) and NHWC views.  src element (n,c,y,x) at
 * src[n*ssn + c*ssc + y*ssh + x*ssw]. */
int wcmc_to_nhwc(const float* src, int64_t ssn, int64_t ssc, int64_t ssh, int64_t ssw,
                 float* dst, int64_t dsn, int64_t dsh, int64_t dsw,
                 int N, int C, int H, int W, void* stream);
int wcmc_from_nhwc(const float* src, int64_t ssn, int64_t ssh, int64_t ssw,
                   float* dst, int64_t dsn, int64_t dsc, int64_t dsh, int64_t dsw,
                   int N, int C, int H, int W, void* stream);

/* ---------------------------------------------------------------- convolution
 * Replaces torch.nn.Conv2d forward/backward inside sbmc.modules.ConvChain
 * (call sites support/networks.py:18-24, train_kpcn.py:213) -- the cuDNN calls
 * enabled at train_kpcn.py:349.
 *
 * Packed weights: wp[n][k], n in [0, Np), k in [0, Kt); k = tap*Kp + ci with
 * tap = ky*ks + kx, Kp = round_up(Cin,4), Kt = round_up(ks*ks*Kp, 32),
 * Np = round_up(Cout,16); padding entries are zero.
 *   mode 0 (forward operand):  wp[co][tap*Kp+ci] = w[co][ci][ky][kx]
 *   mode 1 (data-gradient operand, rows are INPUT channels):
 *                              wp[ci][tap'*Kp'+co] = w[co][ci][ks-1-ky'][ks-1-kx'],
 *                              Kp' = round_up(Cout,4), Np' = round_up(Cin,16)
 */
size_t wcmc_conv2d_packed_elems(int rows, int kchan, int ks);
int wcmc_conv2d_pack_weight(const float* w_oihw, float* wp, int Cout, int Cin, int ks, int mode,
                            void* stream);

/* y = act(conv(x, wp) + bias) [* gate'].  Implicit GEMM on fp32 MFMA.
 *   x: NHWC view (N,H,W,Cin); y: NHWC view (N,Ho,Wo,Cout), Ho = H + 2*pad - ks + 1.
 *   bias: [Cout] or NULL.  act/slope: wcmc_act applied to the result.
 *   gate (optional NHWC view with y's geometry): if non-NULL the result is
 *   multiplied by d act_gate / d pre-activation evaluated from the POST-activation
 *   value stored in gate (1 if gate>0 else gate_slope; gate_act = RELU uses slope 0).
 *   That is the fused ReLU backward used when this call computes a data gradient.
 */
int wcmc_conv2d_igemm(const float* x, int64_t xsn, int64_t xsh, int64_t xsw,
                      int N, int H, int W, int Cin,
                      const float* wp, const float* bias,
                      float* y, int64_t ysn, int64_t ysh, int64_t ysw, int Cout,
                      int ks, int pad, int act, float slope,
                      const float* gate, int64_t gsn, int64_t gsh, int64_t gsw,
                      int gate_act, float gate_slope, void* stream);

/* Weight gradient dW[co][ci][ky][kx] = sum_{n,y,x} dy[n,y,x,co] * x[n,y+ky-pad,x+kx-pad,ci]
 * (+ bias gradient db[co] = sum dy) written in the parameter's own OIHW layout.
 * Two launches inside: split-K partial slabs into `workspace`, then a fixed-order
 * reduction (bitwise reproducible).  db may be NULL. */
size_t wcmc_conv2d_wgrad_workspace_bytes(int N, int Ho, int Wo, int Cout, int Cin, int ks);
int wcmc_conv2d_wgrad(const float* x, int64_t xsn, int64_t xsh, int64_t xsw,
                      int N, int H, int W, int Cin,
                      const float* dy, int64_t dsn, int64_t dsh, int64_t dsw, int Cout,
                      int ks, int pad, float* dw_oihw, float* db,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Strided channel-first (N,C,H,W) fp32 -> dense split tensor in one pass (element strides; C <= 64): the per-sample
 * path descriptors `paths` (support/networks.py:31-33) are only read as the embedding chain's split input. */
int wcmc_split_from_nchw(const float* src, int64_t ssn, int64_t ssc, int64_t ssh, int64_t ssw, void* out_split, int N,
                         int C, int H, int W, void* stream);

/* ---------------------------------------------------------------- split-bf16 ("bf16x3") convolution
 * Same reference expressions as above (torch.nn.Conv2d fwd/bwd in sbmc.modules.ConvChain), computed
 * with every fp32 operand carried as two bf16 planes hi = bf16(x), lo = bf16(x - hi) and each product
 * as hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation (~2^-17 relative per operand;
 * 5.3x the fp32-MFMA rate).  Chain-internal "split tensors" are dense u16 [N][H][W][2][Cp],
 * Cp = round_up(C,8), plane 0 = hi, plane 1 = lo, pad channels zero.  Packed weights:
 * u16 wp[Np][2][Kt], k = tap*round_up(kchan,8) + c, Kt = round_up(ks*ks*Kp, 32), Np = round_up(rows,16);
 * `mode` as for wcmc_conv2d_pack_weight. */
size_t wcmc_split_elems(int N, int H, int W, int C);
int wcmc_split_bf16(const float* x, int64_t xsn, int64_t xsh, int64_t xsw, void* out_split,
                    int N, int H, int W, int C, void* stream);
/* split(dy * act'(post)): the backward of a chain's output activation (wcmc_act_backward) folded into the split of
 * the upstream gradient; dy and post are fp32 NHWC views of the same geometry. */
int wcmc_split_gated_bf16(const float* dy, int64_t dsn, int64_t dsh, int64_t dsw, const float* post, int64_t psn,
                          int64_t psh, int64_t psw, int act, float slope, void* out_split, int N, int H, int W, int C,
                          void* stream);
/* cat([flat (B*S,C1,H,W), repeat_S(prop (B,C2,H,W))], 1) (support/networks.py:39-40) written directly as a
 * split tensor of B*S images with C1 + C2 channels; C1 % 8 == 0. */
int wcmc_cat_broadcast_split(const float* flat, int64_t fsn, int64_t fsh, int64_t fsw,
                             const float* prop, int64_t psn, int64_t psh, int64_t psw,
                             void* out_split, int B, int S, int H, int W, int C1, int C2, void* stream);
/* U-Net skip concatenation (sbmc.modules.Autoencoder: cat([upsample_x2(deeper), skip], 1) feeding a level's right
 * ConvChain): the bilinear upsampling (align_corners=False, as wcmc_upsample2_fwd) is evaluated on the fly and the
 * concatenation is written once, directly as the chain's split input.  deep: (N, C1, H/2, W/2), skip: (N, C2, H, W),
 * C1 % 8 == 0; the result equals wcmc_upsample2_fwd + wcmc_cat_broadcast_split(S = 1) bit for bit. */
int wcmc_cat_upsample_split(const float* deep, int64_t dsn, int64_t dsh, int64_t dsw, const float* skip, int64_t ssn,
                            int64_t ssh, int64_t ssw, void* out_split, int N, int H, int W, int C1, int C2,
                            void* stream);

/* split(g (B*S,C,H,W) + repeat_S(gm (B,C,H,W)) * scale): the gradient of a chain output that feeds both the
 * concatenation and the spp mean (support/networks.py:35-40), as the split dy of the chain's backward.  Either
 * gradient may be null. */
int wcmc_add_broadcast_split(const float* g, int64_t gsn, int64_t gsh, int64_t gsw,
                             const float* gm, int64_t msn, int64_t msh, int64_t msw, float scale,
                             void* out_split, int B, int S, int H, int W, int C, void* stream);
/* The gradient that ENTERS a chain's backward, split, with the column sums of the result (= the last layer's bias
 * gradient, `interfaces.py:237-238` -> `nn.Conv2d` backward) in the same pass:
 *   out = split((dy [+ repeat_S(gm) * scale]) [* act'(post)])      -- dy or gm may be null, post may be null --
 * i.e. wcmc_split_bf16 / wcmc_split_gated_bf16 / wcmc_add_broadcast_split, plus colsum_partial in the layout of
 * wcmc_conv2d_igemm_bf16x3's column sums (wcmc_conv2d_igemm_colsum_elems floats), to be handed to
 * wcmc_conv2d_wgrad_bf16x3 as dy_colsum_partial.  N = B * S images when gm (B images) is given. */
int wcmc_split_dy_colsum_bf16(const float* dy, int64_t dsn, int64_t dsh, int64_t dsw,
                              const float* post, int64_t psn, int64_t psh, int64_t psw, int act, float slope,
                              const float* gm, int64_t msn, int64_t msh, int64_t msw, int S, float scale,
                              void* out_split, float* colsum_partial, int N, int H, int W, int C, void* stream);
/* mode: 0 = forward orientation (rows = Cout, k over Cin x taps); 1 = data-gradient orientation (rows = Cin, k over
 * Cout x flipped taps) for a three-term launch; 2 = the same orientation in the K order of a TWO-term launch
 * (wcmc_conv2d_igemm_bf16x3 with terms = 2: the channel slabs are twice as wide, see there); 3 = the FORWARD orientation in
 * that K order (a forward launch with terms = 2 or 1: round 4, the un-gated output layers of the "bf16x321o" mode); 4 = mode 3
 * with the weights rounded once to fp16 in the hi rows (wcmc_conv2d_out_f16). */
size_t wcmc_conv2d_packed_elems_bf16x3(int rows, int kchan, int ks, int mode);
int wcmc_conv2d_pack_weight_bf16x3(const float* w_oihw, void* wp, int Cout, int Cin, int ks, int mode,
                                   void* stream);
/* The same for up to 20 (layer, mode) pairs of one chain in ONE launch: w[i] OIHW (Cout[i], Cin[i], ks, ks) ->
 * wp[i] (wcmc_conv2d_packed_elems_bf16x3(rows, kchan, ks, mode) u16 each), mode[i] as above; host arrays of n_entries items. */
int wcmc_conv2d_pack_chain_bf16x3(int n_entries, const float* const* w, void* const* wp, const int* Cout,
                                  const int* Cin, const int* mode, int ks, void* stream);
/* Exactly one of y (fp32 NHWC view) and y_split (dense split tensor) receives the result.
 * gate_split (optional, geometry of the output, requires y_split): fused activation-derivative
 * mask evaluated from the hi plane of the post-activation tensor.
 * mask_out (optional, with y_split): uint8 [N*Ho*Wo][round_up(Cout,8)/8], bit c%8 of byte c/8 = (hi plane of
 * output channel c > 0) -- the same predicate at 1/16 of the bytes; gate_mask (optional, instead of gate_split):
 * such a mask of a tensor with the output's geometry.
 * terms: bf16 MFMAs per product -- 3 = W_lo*x_hi + W_hi*x_lo + W_hi*x_hi (the forward); 2 = W_lo*x_hi + W_hi*x_hi, i.e.
 * x rounded to its hi plane (8 mantissa bits), W exact to 16 -- the data gradient of the default mode, whose x operand is
 * dy (wp then packed with mode 2).  Where no two-term kernel instance exists for the shape the launch runs three terms
 * (the packing of mode 2 follows the same rule, so the pair stays consistent).  Replaces the data gradient of
 * `nn.Conv2d` under cuDNN (train_kpcn.py:349; TF32 by default on the reference's hardware: 10 bits on BOTH operands).
 * 1 = W_hi*x_hi only (8 bits on both operands; wp packed with mode 3; round 4): the forward of a 5x5 layer whose output no
 * activation gates -- sbmc.KPCN's kernel-predicting output layers (call site support/interfaces.py:203-204) -- where a
 * rounding cannot flip a ReLU unit; granted where the hi-plane instance of the 64-pixel 5x5 kernel exists (cout blocks of
 * seven tiles, input channels not 24 mod 32), anywhere else the launch runs the plan's two or three terms. */
int wcmc_conv2d_igemm_bf16x3(const void* x_split, int N, int H, int W, int Cin,
                             const void* wp, const float* bias,
                             float* y, int64_t ysn, int64_t ysh, int64_t ysw, void* y_split, int Cout,
                             int ks, int pad, int act, float slope,
                             const void* gate_split, int gate_act, float gate_slope,
                             float* colsum_partial, const void* gate_mask, void* mask_out, int terms, void* stream);
/* Round 4: the forward of an un-gated 5x5 OUTPUT layer with ONE fp16 MFMA per product (the "bf16x321h" mode; sbmc.KPCN's
 * kernel-predicting output layers, call site support/interfaces.py:203-204).  Both operands are rounded ONCE to fp16 (11 bits;
 * the reference's own cuDNN path rounds both to TF32's 10, train_kpcn.py:349): x by wcmc_split_to_f16 (split tensor ->
 * [N*H*W][round_up(C,8)] halfs, saturating), the weights by wcmc_conv2d_pack_weight_bf16x3 with mode 4.  No activation follows
 * the layer, so no ReLU gate can flip (profiles/r04_forward_ladder.txt: every HIDDEN layer needs >= 16-bit operands).
 * wcmc_conv2d_out_f16_supported: 5x5, cout blocks of seven tiles, input channels not 24 mod 32 (the hi-plane instance of the
 * 64-pixel kernel); y = conv(x, W) + bias as an fp32 NHWC view, linear. */
int wcmc_conv2d_out_f16_supported(int Cin, int Cout, int ks);
size_t wcmc_split_to_f16_elems(int N, int H, int W, int C);
int wcmc_split_to_f16(const void* x_split, int N, int H, int W, int C, void* out_f16, void* stream);
int wcmc_conv2d_out_f16(const void* x_f16, int N, int H, int W, int Cin, const void* wp_f16, const float* bias, float* y,
                        int64_t ysn, int64_t ysh, int64_t ysw, int Cout, int ks, int pad, void* stream);
/* colsum_partial (optional, with y_split): [wcmc_conv2d_igemm_colsum_elems] floats that receive the
 * per-pixel-tile column sums of the result -- the bias gradient of the layer that consumes this
 * data gradient, finished by wcmc_colsum_finish (saves a pass over dy per layer).  The buffer ends with a trailer
 * word: the number of rows the producing launch wrote, which is all wcmc_colsum_finish reads. */
/* Two 1x1 layers in one launch (sbmc.modules.ConvChain with ksize 1: the last two layers of PathNet.final and of
 * PathNet.embedding, support/networks.py:22-27,33-41, and the data gradient of the former): y1 = gate * act1(W1 x + b1)
 * is written as a split tensor -- with its 1-bit mask (mask1) on the forward, gated by gate_mask1 and with column
 * sums (colsum1, see below) on the backward, exactly as wcmc_conv2d_igemm_bf16x3 would -- and y2 = act2(W2 y1 + b2)
 * (fp32 NHWC view) is computed from the tile while it is on chip, so the hidden tensor is not re-read by a second
 * launch.  wcmc_conv1x1_pair_supported says whether a fused instance exists for the channel counts (else: two
 * wcmc_conv2d_igemm_bf16x3 launches, same results bit for bit).  wp1 / wp2: wcmc_conv2d_pack_weight_bf16x3. */
int wcmc_conv1x1_pair_supported(int Cin, int Cout1, int Cout2);
int wcmc_conv1x1_pair_bf16x3(const void* x_split, int N, int H, int W, int Cin, const void* wp1, const float* bias1,
                             int Cout1, int act1, float slope1, void* y1_split, void* mask1, const void* gate_mask1,
                             int gate_act1, float gate_slope1, float* colsum1, const void* wp2, const float* bias2,
                             int Cout2, int act2, float slope2, float* y2, int64_t y2sn, int64_t y2sh, int64_t y2sw,
                             void* stream);
/* ---------------------------------------------------------------- PathNet.embedding, fused (support/networks.py:33-36)
 * y = ConvChain(Cin -> 64 -> 64 -> 64, ksize 1, ReLU, ReLU, linear)(x) over M = B*S*H*W pixels as ONE launch per direction:
 * the hidden activations never leave the chip (forward) and are recomputed from x (backward).
 *   x_split   split tensor [M][2][round_up(Cin, 8)] (wcmc_split_from_nchw / wcmc_split_bf16)
 *   wp0..wp2  forward packs (wcmc_conv2d_pack_weight_bf16x3, mode 0), wt1 / wt2 data-gradient packs (mode 1) of layers 1, 2
 *   y         fp32 [M][64] (an NHWC tensor of 64 channels); bit-identical to three wcmc_conv2d_igemm_bf16x3 launches
 *   backward  dy = gy (fp32, pixel stride gy_pixel_stride floats; may be null) + repeat_S(gm) * gm_scale (gm: fp32 over
 *             M / S pixels, B images of HW pixels; may be null) -- the gradient of `y` and of its spp mean
 *             (support/networks.py:35-36) -- and dw / db of the three layers (OIHW, ks = 1) in the default mode's arithmetic
 *             (data gradients dy_hi x (W_hi + W_lo), weight gradients hi x hi); no gradient for x.  Fixed summation order:
 *             bitwise reproducible. */
int wcmc_embed3_supported(int Cin, int C1, int C2, int C3);
size_t wcmc_embed3_bwd_workspace_bytes(void);
int wcmc_embed3_fwd(const void* x_split, int64_t M, int Cin, const void* wp0, const float* b0, const void* wp1,
                    const float* b1, const void* wp2, const float* b2, float* y, void* stream);
/* The same forward with the spp mean of y leaving in the same launch (support/networks.py:35-36: y.view(B, S, ...).mean(1)):
 * y_mean fp32 [M / S][64], the sums formed s ascending in fp32 and multiplied by 1 / S -- bit-identical to wcmc_spp_reduce on
 * the stored y, which is then not read again.  M = B * S * HW with HW % 64 == 0 (wcmc_embed3_mean_supported). */
int wcmc_embed3_mean_supported(int S, int64_t HW);
int wcmc_embed3_mean_fwd(const void* x_split, int64_t M, int Cin, const void* wp0, const float* b0, const void* wp1,
                         const float* b1, const void* wp2, const float* b2, float* y, float* y_mean, int S, int64_t HW,
                         void* stream);
int wcmc_embed3_bwd(const void* x_split, int64_t M, int Cin, const void* wp0, const float* b0, const void* wp1,
                    const float* b1, const void* wt1, const void* wt2, const float* gy, int gy_pixel_stride,
                    const float* gm, int gm_pixel_stride, int S, int64_t HW, float gm_scale, float* dw0, float* db0,
                    float* dw1, float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, void* stream);
/* ---------------------------------------------------------------- PathNet.final, fused (support/networks.py:39-42)
 * out = ConvChain(128 -> 128 -> outc <= 8, ksize 1, ReLU, ReLU)(cat([y, repeat_S(prop)], 1)): y fp32 over M = B*S*HW pixels
 * (64 channels, pixel stride y_pixel_stride floats), prop fp32 over B*HW pixels (64 channels), out / gout fp32 [M][os] with
 * os = 4 for outc <= 4, else 8 (an NHWC view of outc channels; round 4: up to eight, the reference's --pnet_out_size 6 runs,
 * train_kpcn.py:209-212); HW % 64 == 0.  Neither the concatenation nor the hidden activation is written: the backward recomputes
 * them, writes dy [M][64] and dprop [B*HW][64] (the sum over the S samples) and the weight / bias gradients (OIHW, ks = 1) in the
 * default mode's arithmetic.  Forward bit-identical to wcmc_cat_broadcast_split + wcmc_conv1x1_pair_bf16x3.
 *   wp0 / wp1  forward packs (mode 0) of the two layers, wt0 / wt1 their data-gradient packs (mode 1). */
int wcmc_final2_supported(int C1, int C2, int Chid, int outc, int64_t HW);
size_t wcmc_final2_bwd_workspace_bytes(void);
int wcmc_final2_fwd(const float* y, int y_pixel_stride, const float* prop, int prop_pixel_stride, int B, int S, int64_t HW,
                    const void* wp0, const float* b0, const void* wp1, const float* b1, int outc, float* out, void* stream);
int wcmc_final2_bwd(const float* y, int y_pixel_stride, const float* prop, int prop_pixel_stride, int B, int S, int64_t HW,
                    const void* wp0, const float* b0, const void* wp1, const float* b1, int outc, const void* wt0,
                    const void* wt1, const float* gout, float* dy, float* dprop, float* dw0, float* db0, float* dw1,
                    float* db1, void* workspace, size_t workspace_bytes, void* stream);
size_t wcmc_conv2d_igemm_colsum_elems(int N, int Ho, int Wo, int Cout);
int wcmc_colsum_finish(const float* partial, int N, int Ho, int Wo, int Cout, float* db, void* stream);
size_t wcmc_conv2d_wgrad_bf16x3_workspace_bytes(int N, int Ho, int Wo, int Cout, int Cin, int ks);
/* phase: 0 = everything; 1 = the split-K GEMM into the workspace only; 2 = the slab reduction and
 * bias gradient only (1 then 2 == 0; lets a profiler bracket the GEMM launch alone).
 * dy_colsum_partial (optional): the per-tile column sums of dy that the launch which PRODUCED dy left
 * (wcmc_conv2d_igemm_bf16x3 / wcmc_conv1x1_pair_bf16x3 colsum output, wcmc_conv2d_igemm_colsum_elems floats): the bias
 * gradient db is then finished from them by extra blocks of the slab-reduction launch -- same sums, same order as
 * wcmc_colsum_finish, bit for bit -- instead of a column-sum pass over dy plus a finish launch.
 * terms: bf16 MFMAs per product -- 3 = dy_lo*x_hi + dy_hi*x_lo + dy_hi*x_hi; 1 = dy_hi*x_hi only (the lo planes are not
 * read: half the operand bytes).  The reference's counterpart is cuDNN's weight gradient under
 * `torch.backends.cudnn.allow_tf32` (train_kpcn.py:349 leaves the default, TF32 = 10 mantissa bits per operand). */
/* Phase 2 of up to 32 wcmc_conv2d_wgrad_bf16x3 calls in ONE launch: layer i's split-K slabs (workspace[i], filled by a phase-1
 * call with the same N, Ho, Wo, Cout, Cin, ks and terms) are summed into dw[i], and db[i] (optional) is finished from
 * dy_colsum_partial[i] (required with db[i]).  Bit-identical to the per-layer phase 2.  The U-Net's fifteen weight gradients per
 * PathNet and backward pass are reduced by one launch at the end of the chain walk instead of fifteen between its GEMMs. */
int wcmc_conv2d_wgrad_reduce_multi(int n, void* const* workspace, float* const* dw, float* const* db,
                                   const float* const* dy_colsum_partial, const int* N, const int* Ho, const int* Wo,
                                   const int* Cout, const int* Cin, const int* ks, int terms, void* stream);
int wcmc_conv2d_wgrad_bf16x3(const void* x_split, int N, int H, int W, int Cin,
                             const void* dy_split, int Cout, int ks, int pad, float* dw, float* db,
                             void* workspace, size_t workspace_bytes, int phase, const float* dy_colsum_partial,
                             int terms, void* stream);

/* dx = dy * act'(y) from the post-activation value y (NHWC views of equal geometry). */
int wcmc_act_backward(const float* dy, int64_t dsn, int64_t dsh, int64_t dsw,
                      const float* y, int64_t ysn, int64_t ysh, int64_t ysw,
                      float* dx, int64_t xsn, int64_t xsh, int64_t xsw,
                      int N, int H, int W, int C, int act, float slope, void* stream);

/* ---------------------------------------------------------------- kernel apply
 * Replaces sbmc.modules.KernelApply(softmax=True, splat=False) inside sbmc.KPCN
 * (call site support/interfaces.py:203-204; upstream a Halide op).
 *   logits: NHWC view (N,h,w,k*k) -- tap t = (dy+r)*k + (dx+r), r = k/2.
 *   data/out: (N,C,h,w) with arbitrary element strides (sn,sc,sh,sw), C <= 4.
 *   out[n,c,y,x] = sum_t softmax_t(logits[n,y,x,:]) * data0[n,c,y+dy,x+dx],
 *   data0 = data zero-extended outside [0,h)x[0,w).
 *   lse: [N*h*w] per-pixel log-sum-exp saved for the backward (may be NULL).
 */
int wcmc_kernel_apply_fwd(const float* logits, int64_t lsn, int64_t lsh, int64_t lsw,
                          const float* data, int64_t dsn, int64_t dsc, int64_t dsh, int64_t dsw,
                          float* out, int64_t osn, int64_t osc, int64_t osh, int64_t osw,
                          float* lse, int N, int C, int h, int w, int k, void* stream);
/* d_logits (NHWC view, same geometry as logits) from grad_out; d_data (N,C,h,w
 * contiguous, accumulated with atomics, must be zeroed by the caller) may be NULL. */
int wcmc_kernel_apply_bwd(const float* logits, int64_t lsn, int64_t lsh, int64_t lsw,
                          const float* data, int64_t dsn, int64_t dsc, int64_t dsh, int64_t dsw,
                          const float* out, int64_t osn, int64_t osc, int64_t osh, int64_t osw,
                          const float* grad_out, int64_t gsn, int64_t gsc, int64_t gsh, int64_t gsw,
                          const float* lse,
                          float* d_logits, int64_t qsn, int64_t qsh, int64_t qsw,
                          float* d_data, int N, int C, int h, int w, int k, void* stream);

/* Tail of sbmc.KPCN.forward (result keys consumed at support/interfaces.py:207-211):
 *   radiance = albedo * r_diffuse + exp(r_specular) - 1.
 * Inputs (N,C,H,W) with arbitrary element strides; out / grad_out / d_* contiguous (N,C,H,W). */
int wcmc_recombine_fwd(const float* albedo, int64_t asn, int64_t asc, int64_t ash, int64_t asw,
                       const float* r_diffuse, int64_t dsn, int64_t dsc, int64_t dsh, int64_t dsw,
                       const float* r_specular, int64_t ssn, int64_t ssc, int64_t ssh, int64_t ssw,
                       float* out, int N, int C, int H, int W, void* stream);
int wcmc_recombine_bwd(const float* grad_out,
                       const float* albedo, int64_t asn, int64_t asc, int64_t ash, int64_t asw,
                       const float* r_specular, int64_t ssn, int64_t ssc, int64_t ssh, int64_t ssw,
                       float* d_diffuse, float* d_specular, int N, int C, int H, int W, void* stream);

/* Image losses on the (N,C,H,W) outputs: torch.nn.L1Loss (train_kpcn.py:299-304; applied at
 * support/interfaces.py:213-249) and RelativeMSE (support/losses.py:245-264: 0.5 * mean((x - ref)^2 / (ref^2 + eps))) of
 * one pair in one pass; either output may be null.  Both tensors with arbitrary element strides.  Deterministic
 * (fixed-order two-level sum).  wcmc_l1_mean_bwd: dx = grad_loss[0] * sign(x - ref) / (N*C*H*W), contiguous (N,C,H,W). */
size_t wcmc_image_loss_workspace_bytes(void);
int wcmc_image_loss_fwd(const float* x, int64_t xsn, int64_t xsc, int64_t xsh, int64_t xsw,
                        const float* ref, int64_t rsn, int64_t rsc, int64_t rsh, int64_t rsw, float eps,
                        float* l1_mean, float* relative_mse, void* workspace, size_t workspace_bytes,
                        int N, int C, int H, int W, void* stream);
int wcmc_l1_mean_bwd(const float* x, int64_t xsn, int64_t xsc, int64_t xsh, int64_t xsw,
                     const float* ref, int64_t rsn, int64_t rsc, int64_t rsh, int64_t rsw,
                     const float* grad_loss, float* dx, int N, int C, int H, int W, void* stream);

/* ---------------------------------------------------------------- U-Net glue
 * F.max_pool2d(x,2,2) / F.interpolate(x, scale_factor=2, 'bilinear',
 * align_corners=False) inside sbmc.modules.Autoencoder (support/networks.py:20-22). */
int wcmc_maxpool2_fwd(const float* x, int64_t xsn, int64_t xsh, int64_t xsw,
                      float* y, int64_t ysn, int64_t ysh, int64_t ysw,
                      int N, int H, int W, int C, void* stream);
/* dx[n,2y+i,2x+j] = dy[n,y,x] where x[...] is the (first) maximum of its window, else 0. */
/* dx = maxpool2's gradient + add: the pooled tensor also feeds a skip connection, whose gradient `add` (fp32 NHWC view of x's
 * geometry) is summed in by the same pass (the sum autograd would form with one more elementwise launch). */
int wcmc_maxpool2_bwd_add(const float* x, int64_t xsn, int64_t xsh, int64_t xsw, const float* dy, int64_t dsn, int64_t dsh,
                          int64_t dsw, const float* add, int64_t asn, int64_t ash, int64_t asw, float* dx, int64_t gsn,
                          int64_t gsh, int64_t gsw, int N, int H, int W, int C, void* stream);
int wcmc_maxpool2_bwd(const float* x, int64_t xsn, int64_t xsh, int64_t xsw,
                      const float* dy, int64_t dsn, int64_t dsh, int64_t dsw,
                      float* dx, int64_t gsn, int64_t gsh, int64_t gsw,
                      int N, int H, int W, int C, void* stream);
/* (N,H,W,C) -> (N,2H,2W,C) */
int wcmc_upsample2_fwd(const float* x, int64_t xsn, int64_t xsh, int64_t xsw,
                       float* y, int64_t ysn, int64_t ysh, int64_t ysw,
                       int N, int H, int W, int C, void* stream);
/* dy (N,2H,2W,C) -> dx (N,H,W,C) (gather form of the transposed interpolation) */
int wcmc_upsample2_bwd(const float* dy, int64_t dsn, int64_t dsh, int64_t dsw,
                       float* dx, int64_t xsn, int64_t xsh, int64_t xsw,
                       int N, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------- PathNet glue
 * support/networks.py:35-36 (`flat.mean(1)`) and :39-40 (`repeat` + `cat`).
 * x holds B*S images, y holds B images (image b*S+s belongs to patch b).
 *   reduce:    y[b] = scale * sum_s x[b*S+s]
 *   broadcast: y[b*S+s] = (accumulate ? y[b*S+s] : 0) + scale * x[b]            */
int wcmc_spp_reduce(const float* x, int64_t xsn, int64_t xsh, int64_t xsw,
                    float* y, int64_t ysn, int64_t ysh, int64_t ysw,
                    int B, int S, int H, int W, int C, float scale, void* stream);
int wcmc_spp_broadcast(const float* x, int64_t xsn, int64_t xsh, int64_t xsw,
                       float* y, int64_t ysn, int64_t ysh, int64_t ysw,
                       int B, int S, int H, int W, int C, float scale, int accumulate,
                       void* stream);

/* Per-sample feature assembly of the sample-based denoisers (SBMCInterface / LBMCInterface,
 * support/interfaces.py:394-403, :797-806): out (B, S, C + Cp + 1, H, W, contiguous) =
 * cat([features (B,S,C,H,W), P (B,S,Cp,H,W), repeat_S(P.var(1).mean(1, keepdims) / S)], 2); element strides for the
 * two inputs.  The variance channel carries no gradient (.detach() in the reference): the backward is two slices. */
int wcmc_sample_cat_fwd(const float* feat, int64_t fsb, int64_t fss, int64_t fsc, int64_t fsh, int64_t fsw,
                        const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh, int64_t psw, float* out,
                        int B, int S, int C, int Cp, int H, int W, void* stream);

/* ---------------------------------------------------------------- P-buffer statistics
 * support/interfaces.py:165-176: builds the KPCN input
 *   out = cat([base, mean_s P, (var_s P (unbiased)).mean_c / S], channel)
 * base: (B,Cb,H,W) strided NCHW-style tensor; P: (B,S,Cp,H,W) strided;
 * out: NHWC view (B,H,W,Cb+Cp+1).  The backward of the mean term is
 *   dP[b,s,c,y,x] = g[b,y,x,Cb+c] / S   (the variance term is detached, :165).   */
int wcmc_pbuffer_cat_fwd(const float* base, int64_t bsn, int64_t bsc, int64_t bsh, int64_t bsw,
                         const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh,
                         int64_t psw,
                         float* out, int64_t osn, int64_t osh, int64_t osw,
                         int B, int S, int Cb, int Cp, int H, int W, void* stream);
int wcmc_pbuffer_cat_bwd(const float* g, int64_t gsn, int64_t gsh, int64_t gsw,
                         float* dp, int64_t psb, int64_t pss, int64_t psc, int64_t psh,
                         int64_t psw,
                         int B, int S, int Cb, int Cp, int H, int W, void* stream);

/* ---------------------------------------------------------------- FeatureMSE
 * support/losses.py:33-61,63-65,82-113 (path-disentangling loss, color='rgb').
 * p: (B,S,C,h,w) strided view (already cropped), C <= 8; ref: (B,3,h,w) strided
 * (NOT yet tonemapped).  idx_patch: permutation of S*h*w (int64, device);
 * idx_batch: permutation of B*S*h*w or NULL for non_local=False.
 * workspace layout is private; loss is a single float.  The forward leaves in the
 * workspace what the backward needs (tonemapped ref, displacements, inverse
 * permutations), so the same workspace must be passed to the backward.          */
size_t wcmc_feature_mse_workspace_bytes(int B, int S, int C, int h, int w);
int wcmc_feature_mse_fwd(const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh,
                         int64_t psw,
                         const float* ref, int64_t rsb, int64_t rsc, int64_t rsh, int64_t rsw,
                         const int64_t* idx_patch, const int64_t* idx_batch,
                         float* loss, void* workspace, size_t workspace_bytes,
                         int B, int S, int C, int h, int w, void* stream);
/* dp: contiguous (B,S,C,h,w); grad_scale: device pointer to the upstream scalar gradient. */
int wcmc_feature_mse_bwd(const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh,
                         int64_t psw,
                         const int64_t* idx_patch, const int64_t* idx_batch,
                         const float* grad_scale, float* dp,
                         void* workspace, size_t workspace_bytes,
                         int B, int S, int C, int h, int w, void* stream);

/* GlobalRelativeSimilarityLoss (support/losses.py:116-211, `--manif_loss GRS`): same pairings and
 * displacements, loss = (logsumexp(alpha*[d_p, d_b, -d_p, -d_b, 0]) - log(1 + 4N)) / sqrt(alpha).
 * Same workspace (wcmc_feature_mse_workspace_bytes) and calling convention as FeatureMSE. */
int wcmc_grs_fwd(const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh, int64_t psw,
                 const float* ref, int64_t rsb, int64_t rsc, int64_t rsh, int64_t rsw,
                 const int64_t* idx_patch, const int64_t* idx_batch, float alpha,
                 float* loss, void* workspace, size_t workspace_bytes,
                 int B, int S, int C, int h, int w, void* stream);
int wcmc_grs_bwd(const float* p, int64_t psb, int64_t pss, int64_t psc, int64_t psh, int64_t psw,
                 const int64_t* idx_patch, const int64_t* idx_batch,
                 const float* grad_scale, float* dp,
                 void* workspace, size_t workspace_bytes,
                 int B, int S, int C, int h, int w, void* stream);

/* A pseudo-random permutation of [0, n) as int64 indices, keyed by `seed` (6-round Feistel network, cycle-walked):
 * the `rng='device'` source of the FeatureMSE / GRS pairings instead of the sort behind torch.randperm.  The
 * reference draws its pairings with torch.randperm on the CPU generator (support/losses.py:35,50). */
int wcmc_random_permutation(int64_t* out, int64_t n, uint64_t seed, void* stream);
/* The same bijection keyed from DEVICE memory, for a launch captured into the step's hipGraph (a by-value seed is frozen at capture):
 * state = {seed, step counter} (two uint64), key = wcmc_permutation_key(seed, counter, slot) (host mirror of the device arithmetic:
 * wcmc_random_permutation_dev(out, n, state, slot) == wcmc_random_permutation(out, n, wcmc_permutation_key(state[0], state[1], slot))).
 * wcmc_step_counter_advance: state[1] += 1, once per step, ahead of the step's draws (losses.py:35,50 draws fresh pairings per call).
 * slot in [0, 8): the step's draws (diffuse patch / batch, specular patch / batch). */
uint64_t wcmc_permutation_key(uint64_t seed, uint64_t counter, int slot);
int wcmc_random_permutation_dev(int64_t* out, int64_t n, const uint64_t* state, int slot, void* stream);
int wcmc_step_counter_advance(uint64_t* state, void* stream);

/* ---------------------------------------------------------------- image losses of the sample-based interfaces
 * support/losses.py:267-320, built at train_lbmc.py:164-170 / train_sbmc.py (SMAPE: LBMC; Tonemapped*: SBMC).  Strided (N,C,H,W)
 * operands like wcmc_image_loss_fwd; one pass + a one-block finish in a fixed order (bitwise reproducible); workspace of
 * wcmc_image_loss_workspace_bytes().  T = Reinhard tone map of the clamped image (losses.py:234-242).
 *   kind 0  SMAPE                  mean |x - ref| / (eps + |x| + |ref|)   (the denominator carries no gradient, losses.py:279-282)
 *   kind 1  TonemappedMSE          0.5 * mean (T(x) - T(ref))^2
 *   kind 2  TonemappedRelativeMSE  0.5 * mean (T(x) - T(ref))^2 / (T(ref)^2 + eps)
 * bwd: dx (contiguous (N,C,H,W)) = *grad_loss * d loss / d x. */
int wcmc_image_loss2_fwd(int kind, const float* x, int64_t xsn, int64_t xsc, int64_t xsh, int64_t xsw, const float* ref,
                         int64_t rsn, int64_t rsc, int64_t rsh, int64_t rsw, float eps, float* loss, void* workspace,
                         size_t workspace_bytes, int N, int C, int H, int W, void* stream);
int wcmc_image_loss2_bwd(int kind, const float* x, int64_t xsn, int64_t xsc, int64_t xsh, int64_t xsw, const float* ref,
                         int64_t rsn, int64_t rsc, int64_t rsh, int64_t rsw, float eps, const float* grad_loss, float* dx,
                         int N, int C, int H, int W, void* stream);
/* torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) (support/interfaces.py:454-458, 826-833: 1000 for SBMC, 250 for LBMC)
 * over up to 96 gradient tensors in three launches: norm_and_coef[0] = the total 2-norm BEFORE clipping (what the reference prints),
 * norm_and_coef[1] = min(1, max_norm / (norm + 1e-6)); the gradients are scaled in place when the factor is below one. */
size_t wcmc_grad_norm_clip_workspace_bytes(int n_tensors, const int64_t* numel);
int wcmc_grad_norm_clip(int n_tensors, float* const* grads, const int64_t* numel, float max_norm, float* norm_and_coef,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- weight normalisation
 * sbmc.modules.ConvChain wraps every nn.Conv2d in torch.nn.utils.weight_norm unless its caller passes weight_norm=False;
 * support/networks.py:18-24 (PathNet's embedding / propagation / final chains) does not, sbmc.KPCN does.  Parameters per
 * layer: weight_g (Cout,1,1,1), weight_v (Cout,Cin,k,k);  weight = weight_g * weight_v / ||weight_v||, norm over (Cin,k,k)
 * per output channel (torch._weight_norm, dim 0).
 *
 * Both entries take ALL layers of a model in one launch (n_layers <= 32): tables of n_layers device pointers, rows[l] = Cout,
 * row_len[l] = Cin*k*k.  v / w / dw / dv are dense [rows][row_len] fp32 (16-byte aligned when row_len % 4 == 0);
 * g / norm / dg are [rows].
 *   fwd: w = v * (g / ||v||), norm = ||v||                                  (torch._weight_norm)
 *   bwd: dg = <dw, v> / norm;  dv = (g / norm) * (dw - v * <dw, v> / norm^2)   (torch._weight_norm_interface_backward) */
int wcmc_weight_norm_fwd(int n_layers, const float* const* v, const float* const* g, float* const* w,
                         float* const* norm, const int* rows, const int* row_len, void* stream);
int wcmc_weight_norm_bwd(int n_layers, const float* const* dw, const float* const* v, const float* const* g,
                         const float* const* norm, float* const* dv, float* const* dg, const int* rows,
                         const int* row_len, void* stream);

/* ---------------------------------------------------------------- clip + Adam
 * support/interfaces.py:260-261 (clip_grad_value_) + :269-271 (Adam.step,
 * train_kpcn.py:274-277: default betas/eps, no weight decay, no amsgrad) fused
 * over one flat parameter buffer.  grad is clamped IN PLACE (the reference leaves
 * clipped .grad behind), then m,v,param are updated.  step is the 1-based count. */
/* guard (optional device float): when *guard == 0 the launch is a no-op (the host raises the
 * reference's non-finite-loss error, interfaces.py:254-257, without having to sync before enqueuing). */
int wcmc_clip_adam(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float clip, double lr, double beta1, double beta2, double eps, int step,
                   float grad_scale, const float* guard, void* stream);
/* The same launch with its per-step scalars read from DEVICE memory, for a launch captured into the step's hipGraph (whose
 * by-value arguments are frozen at capture): hyper7 = the seven floats wcmc_clip_adam_hyper (host-side, no GPU call)
 * derives from (lr, betas, eps, step) exactly as wcmc_clip_adam does -- same arithmetic, bit-identical update.  The host
 * refreshes the buffer before every replay (`optim.param_groups[0]['lr']` may have changed: train_kpcn.py:279-296). */
void wcmc_clip_adam_hyper(double lr, double beta1, double beta2, double eps, int step, float* out7);
int wcmc_clip_adam_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float clip,
                       float grad_scale, const float* hyper7, const float* guard, void* stream);
/* The head of the step's captured tail in one launch: losses = n (<= 16) device pointers to the 0-d loss scalars of loss_dict.
 * flags[i] = isfinite(loss_i) (the reference's check, support/interfaces.py:254-257), flags[n] = guard = all finite AND *ok;
 * *ok <- guard (a non-finite step keeps the steps enqueued behind it from updating until the host has raised the error and reset
 * *ok to 1); sums[i] += loss_i when the guard holds (the running sums of interfaces.py:263-267).  flags + n is what
 * wcmc_clip_adam_dev takes as its guard. */
int wcmc_step_guard(const float* const* losses, int n, float* ok, float* sums, float* flags, void* stream);
/* The same in two halves for several ranks (the reference's nn.DataParallel, train_kpcn.py:256-271, sees one process; here every rank
 * checks its own losses and all must agree): `local` writes flags[0..n) and this rank's 1 - (all finite AND *ok) into flag_slot -- the
 * float behind the first gradient bucket, summed over the ranks by the bucket's all-reduce; `global` reads the summed slot: guard =
 * (slot == 0) -> flags[n], *ok, and sums[i] += loss_i under the guard. */
int wcmc_step_guard_local(const float* const* losses, int n, const float* ok, float* flags, float* flag_slot, void* stream);
int wcmc_step_guard_global(const float* const* losses, int n, const float* flag_slot, float* ok, float* sums, float* flags, void* stream);

/* ---------------------------------------------------------------- per-image preprocessing (data step before the path)
 * support/datasets.py: DenoiseDataset._preprocess_llpm :302-361, ._preprocess_kpcn :487-582,
 * ._gradients :286-300; raw channel map :223-267 (C >= 38 + 11*(max_depth+1); the reference's MAX_DEPTH is 5,
 * C = 104).  Dense, contiguous fp32 device buffers in the reference's numpy layouts:
 *   raw (h, w, s, C);  llpm out (h, w, s, 7 + 5*(max_depth+1)) = 37;  kpcn out (h, w, 44);
 *   gradients: buf (h, w, c) -> out (h, w, 2c) = [d/dx (c), d/dy (c)], zero first column / row. */
int wcmc_preprocess_llpm(const float* raw, int64_t nsamples /* h*w*s */, int C, int max_depth, float* out,
                         void* stream);
size_t wcmc_preprocess_kpcn_workspace_bytes(int h, int w);
int wcmc_preprocess_kpcn(const float* raw, int h, int w, int s, int C, int max_depth, float* out,
                         void* workspace, size_t workspace_bytes, void* stream);
int wcmc_gradients(const float* buf, int h, int w, int c, float* out, void* stream);

/* Batch assembly for the KPCN base model (DenoiseDataset.__getitem__ + _sample_patches + _transpose,
 * support/datasets.py:795-840,1026-1146): crops B windows of P x P pixels at origins[b] = (row, column) out of the
 * per-image buffers kpcn (H, W, 44), llpm (H, W, S, 37; null without --use_llpm_buf) and gt (H, W, 9) and writes the
 * batch dictionary's tensors channel-first and contiguous: diffuse_in / specular_in (B, 34 [+1], P, P), the two
 * 3-channel radiance buffers, albedo + 0.00316, paths (B, S, 36, P, P), and the three targets
 * (total, diffuse / (albedo + 0.00316), log(1 + total - diffuse)).  origins: device int32 [B][2], windows in bounds. */
int wcmc_assemble_kpcn_patches(const float* kpcn, const float* llpm, const float* gt, const int* origins, int B, int H,
                               int W, int S, int P, float* diffuse_in, float* specular_in, float* diffuse_buffer,
                               float* specular_buffer, float* albedo, float* paths, float* target_diffuse,
                               float* target_specular, float* target_total, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WCMC_HIP_H */
