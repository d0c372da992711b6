/*
 * libwesup_hip.so -- C ABI of the MI355X (gfx950) WESUP training-step hot path.
 *
 * Drop-in boundary: the reference is pure Python on stock
 * ATen ops (no native layer), so each entry below replaces the ATen op
 * sequence at the cited reference lines.  A maintainer binds them with ctypes
 * (see INTEGRATION.md); no torch types appear in any signature.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host";
 *   - activations are NHWC fp32 ([B][H][W][C], C contiguous), index tensors int32;
 *   - `stream` is a hipStream_t passed as void*; entries never synchronise,
 *     never allocate, never throw; scratch comes in through (ws, ws_bytes) with
 *     a matching *_workspace_bytes() query;
 *   - return 0 (WESUP_OK) or a negative WESUP_ERR_* code (wesup_strerror()).
 */
#ifndef WESUP_HIP_H
#define WESUP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WESUP_OK 0
#define WESUP_ERR_INVALID (-1)   /* bad shape / null pointer / unsupported size */
#define WESUP_ERR_LAUNCH (-2)    /* hipLaunch failed */
#define WESUP_ERR_WORKSPACE (-3) /* workspace too small */

/* flags for wesup_gemm_nt */
#define WESUP_RELU_IN 1    /* A := max(A, 0) while loading */
#define WESUP_RELU_OUT 2   /* C := max(C, 0) */
#define WESUP_ACCUM 4      /* C += result */
#define WESUP_MASK 8       /* result := mask > 0 ? result : 0 (ReLU backward) */

int wesup_abi_version(void);   /* 6 (round 6: the tile-form pooling entries of ABI 5, wesup_sp_tiles / wesup_sp_pool_tiles_*, are gone) */
/* The two debug entries only work in the debug build of the library (make -C wesup_amd/csrc debug ->
 * libwesup_hip_debug.so, GEMM kernels compiled with the in-kernel clock probe and the per-block trace); the shipped
 * library's kernels carry neither and both entries return WESUP_ERR_INVALID there.
 * debug, host-synchronous: shader clock (MHz) held during the main loop of the last wesup_gemm_nt / conv3x3 fwd /
 * dgrad launch (s_memtime / s_memrealtime of block 0) */
int wesup_debug_clock(double* mhz_out /* host */);
/* debug, host-synchronous: per-block {start, loop start, loop end, end} timestamps (100 MHz ticks) of NT launches */
int wesup_debug_set_trace(void* device_buf);
const char* wesup_strerror(int code);

/* ------------------------------------------------------------------ layout packing
 * models/wesup.py:263-279 feeds NCHW (1,3,H,W); the kernels want NHWC with the
 * 3 image channels padded to 4.  Conv weights stay in torch's (Co,Ci,3,3) layout
 * in the state_dict; they are re-packed per step. */
int wesup_pack_input(const float* img_nchw, float* out_nhwc4, int B, int H, int W, void* stream);
/* w_fwd  [Co][Kf], Kf = wesup_conv3x3_kpad(Ci): k = ((ci/32)*9 + kh*3+kw)*32 + ci%32 for Ci >= 32 (32-channel chunk,
 *         tap, channel: the taps of a chunk are consecutive K-steps); image layer: k = (kh*3+kw)*4 + ci, zero padded
 * w_dgrad[Ci][9*Co]: k = ((co/32)*9 + (2-kh)*3+(2-kw))*32 + co%32          (either panel may be NULL) */
int wesup_conv3x3_kpad(int Ci);
int wesup_pack_conv3x3_weight(const float* w_kcrs, float* w_fwd, float* w_dgrad, int Co, int Ci, void* stream);
int wesup_transpose(const float* in, float* out, int rows, int cols, void* stream);
/* n <= 40 transposes in one launch (the weight panels the input-gradient GEMMs of the side convs / fc layers read,
 * models/wesup.py:213-232; the interpolation-pooling matrices of a batch); items is a HOST array */
typedef struct WesupTransposeItem { const float* in; float* out; int rows, cols; } WesupTransposeItem;
int wesup_transpose_batched(const WesupTransposeItem* items /* host */, int n, void* stream);

/* ------------------------------------------------------------------ input pipeline
 * replaces the per-item CPU augmentation (albumentations) of utils/data.py:116-133,302-327 for a batch of decoded,
 * resized uint8 images: flips + shift/scale/rotate as ONE inverse affine map per image (bilinear image, nearest mask,
 * reflect-101 borders), HueSaturationValue, RandomBrightnessContrast, ToTensor.  params: 12 floats per image
 * {a00,a01,a02,a10,a11,a12 (output pixel -> source), alpha, beta, hue, sat, val, 0}.  mask_hw / out_mask may be NULL.
 * out_img fp32 [B][3][H][W] in [0,1]; out_mask uint8 one-hot [B][C][H][W] (class index 255 = no class).
 * elastic_field (optional, NULL = none): ElasticTransform's displacement field (utils/data.py:124, albumentations
 * ElasticTransform(alpha=1, sigma=50): gaussian_filter(U(-1,1), sigma) * alpha per axis), as a coarse grid [B][2][hc][wc]
 * (dx plane, dy plane; one value per cell x cell pixels, already smoothed -- wesup_amd.utils.data.elastic_field), sampled
 * bilinearly; elastic_params: 12 floats per image {e00,e01,e02,e10,e11,e12 (source grid -> the grid the field is defined
 * on = the transform's own forward affine), l00,l01,l10,l11 (linear part of its inverse), on (0/1), 0}. */
int wesup_augment(const uint8_t* img_hwc, const uint8_t* mask_hw, const float* params, const float* elastic_field,
                  const float* elastic_params, int hc, int wc, int cell, float* out_img_nchw,
                  uint8_t* out_mask_chw, int B, int H, int W, int C, void* stream);
/* Appearance transforms that need a neighbourhood, on the un-warped uint8 images and in the reference's order
 * (utils/data.py:119-125, 306-312): HueSaturationValue -> RandomBrightnessContrast -> CLAHE (8x8 tiles on the L channel
 * of 8-bit Lab, OpenCV's algorithm) -> Blur (3x3 box), every stage rounding to uint8.  params: 8 floats per image
 * {alpha, beta, hue, sat, val, clahe_clip (0 = off), blur (0/1), 0}.  img, out: [B][H][W][3] uint8 (out != img).
 * Run wesup_augment afterwards with neutral colour parameters for the geometry, ToTensor and the one-hot mask. */
size_t wesup_appearance_workspace_bytes(int B, int H, int W);
int wesup_appearance(const uint8_t* img_hwc, const float* params, uint8_t* out_hwc, int B, int H, int W,
                     void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ VGG16 3x3 convs (K1/K2/K12)
 * replaces torchvision VGG16 Conv2d(k=3,pad=1)+ReLU (models/wesup.py:199,279) and its autograd.
 * y is the PRE-ReLU output (the hook taps it, models/wesup.py:246-253); the next
 * layer applies ReLU while loading (relu_in).  x has Cin channels (4 for the image). */
/* ws (optional, may be NULL): room for the partial accumulator tiles of the stream-K blocks that balance a short last
 * round of tiles over all block slots; without it the plain tiling is used.  Size for (Cin -> Cout); for dgrad ask
 * with the channel counts swapped.  One workspace per stream that may run concurrently. */
size_t wesup_conv3x3_workspace_bytes(int B, int H, int W, int Cin, int Cout);
/* y_relu (optional, may be NULL): a second output max(y, 0).  A layer's ReLU'd output is read 9 taps x (Cout/128)
 * times by the next layer's forward and again by its wgrad; applying the ReLU once where the value is produced instead
 * of on every load takes 32 vector instructions per 64 MFMAs out of both loops (relu_in = 0 on the ReLU'd copy). */
int wesup_conv3x3_fwd(const float* x, const float* w_fwd, const float* bias, float* y, float* y_relu,
                      int B, int H, int W, int Cin, int Cout, int relu_in,
                      void* ws, size_t ws_bytes, void* stream);
/* The same forward for a layer with Cout in {64, 128}, with the layer's 1x1 side conv (models/wesup.py:256-266:
 * Conv2d(Cout, Cout/2, 1) on the hooked conv output) fused into the epilogue: side_out[pixel*ld_side + c] =
 * sum_k y[pixel][k] * side_w[c][k] + side_bias[c] for c < Cout/2, computed from the output tile while it is in LDS
 * (y is not read a second time).  side_w row-major (Cout/2, Cout) = the Conv2d weight; side_bias may be NULL. */
int wesup_conv3x3_fwd_side(const float* x, const float* w_fwd, const float* bias, float* y, float* y_relu,
                           const float* side_w, const float* side_bias, float* side_out, int ld_side,
                           int B, int H, int W, int Cin, int Cout, int relu_in, void* stream);
/* dx = conv_transpose(dy) ; if mask_src: dx = mask_src > 0 ? dx : 0 ; if accumulate: dx += old dx */
int wesup_conv3x3_dgrad(const float* dy, const float* w_dgrad, const float* mask_src, float* dx,
                        int B, int H, int W, int Cin, int Cout, int accumulate,
                        void* ws, size_t ws_bytes, void* stream);
/* dw in torch layout (Co,Ci,3,3); db (Co).  Ci is the TRUE channel count (3 for the image,
 * whose tensor has 4); relu_in applies ReLU to x while loading. */
size_t wesup_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Ci, int Cout);
int wesup_conv3x3_wgrad(const float* x, const float* dy, float* dw_kcrs, float* db,
                        int B, int H, int W, int Ci, int Cout, int relu_in,
                        void* ws, size_t ws_bytes, void* stream);
/* The same dw / db through the Winograd F(m x m, 3x3) domain, m = 2 or 4 (the scheme of the non-fused Winograd
 * backward-filter algorithms of vendor conv libraries; it replaces autograd of the reference's Conv2d(k=3,pad=1),
 * models/wesup.py:199): both tensors are transformed per m x m output tile (workspace: P x tiles x (Ci + Cout) floats +
 * split-K slabs, P = (m+2)^2 positions, tiles = B * ceil(H/m) * ceil(W/m)), P TN GEMMs with K = tiles accumulate the
 * transformed filter gradient, and G^T (.) G maps it back to 3x3.  2.25x (m = 2) / 4x (m = 4) fewer multiply-adds than
 * the direct form; same result up to fp32 rounding (1-3e-6 of the tensor's maximum, the direct kernels' own level: m = 4
 * uses the interpolation points 0, +-3/4, +-3/2, inf, ~4x more accurate in fp32 than the textbook 0, +-1, +-2, inf).  Ci, Cout multiples of 4, >= 32 (meant for the 128..512-channel layers); x has Ci
 * channels. */
size_t wesup_conv3x3_wgrad_winograd_workspace_bytes(int B, int H, int W, int Ci, int Cout, int m);
/* v_pre (optional): the transformed input [P][tiles][Ci] kept by wesup_conv3x3_fwd_winograd of the same layer and the
 * same m; then x may be NULL and its transform pass is skipped. */
int wesup_conv3x3_wgrad_winograd(const float* x, const float* v_pre, const float* dy, float* dw_kcrs, float* db,
                                 int B, int H, int W, int Ci, int Cout, int relu_in, int m,
                                 void* ws, size_t ws_bytes, void* stream);
/* Forward and input gradient of the deep layers in the same domain:  V = B^T d B per (m+2) x (m+2) input patch, P batched
 * NT GEMMs M_p = V_p . U_p^T (one launch), Y = A^T M A + the epilogue of wesup_conv3x3_fwd / _dgrad (bias, second ReLU'd
 * output / ReLU mask, accumulate).  U = G g G^T comes from wesup_winograd_pack_weight (once per step):
 * u_fwd [P][Cout][Cin], u_dgrad [P][Cin][Cout] (rotated filter), P*Cin*Cout floats each.  Workspace: transformed
 * input + transformed output, P x tiles x (Cin + Cout) floats (dgrad: ask with the channel counts swapped).
 * v_keep (optional): where the forward leaves V for the weight gradient (P x tiles x Cin floats).
 * y_pool (optional): a third output (B, H/2, W/2, Cout) = the 2x2 / stride-2 max-pool that follows the layer
 * (models/wesup.py:199, torchvision's MaxPool2d(2, 2); an m x m output tile holds (m/2)^2 pooling windows), ReLU'd if
 * pool_relu.  Cin % 32 == 0, Cout % 4 == 0; the results equal the direct kernels' up to fp32 rounding (see above). */
size_t wesup_winograd_weight_floats(int Cin, int Cout, int m);
long wesup_winograd_tiles(int B, int H, int W, int m);
int wesup_winograd_pack_weight(const float* w_kcrs, float* u_fwd, float* u_dgrad, int Cout, int Cin, int m, void* stream);
/* the F(4x4,3x3) filters of several layers in ONE launch (what n calls of wesup_winograd_pack_weight(..., m = 4) write; the
 * step re-derives 12 + 12 filter sets per iteration); layers is a HOST array, at most 32 non-NULL panels in all */
typedef struct WesupWinoFilter { const float* w; float* u_fwd; float* u_dgrad; int Cout, Cin; } WesupWinoFilter;
int wesup_winograd_pack_weights(const WesupWinoFilter* layers /* host */, int n, void* stream);
size_t wesup_conv3x3_winograd_workspace_bytes(int B, int H, int W, int Cin, int Cout, int m);
int wesup_conv3x3_fwd_winograd(const float* x, const float* u_fwd, const float* bias, float* y, float* y_relu,
                               float* y_pool, int pool_relu, float* v_keep,
                               int B, int H, int W, int Cin, int Cout, int relu_in, int m,
                               void* ws, size_t ws_bytes, void* stream);
int wesup_conv3x3_dgrad_winograd(const float* dy, const float* u_dgrad, const float* mask_src, float* dx,
                                 int B, int H, int W, int Cin, int Cout, int accumulate, int m,
                                 void* ws, size_t ws_bytes, void* stream);
/* The input gradient of a layer that FOLLOWS a max-pool, taken straight through the pooling's backward (autograd of
 * ReLU -> MaxPool2d(2,2), models/wesup.py:199; m = 4 only): dy (B,H,W,Cout) at pooled resolution; nothing is written at
 * that resolution -- every value is added to unpool_dst (B,Hu,Wu,Cin), the gradient w.r.t. the pre-pool activations
 * unpool_src (same shape; H == Hu/2, W == Wu/2), at the FIRST maximum of its 2x2 window (torch's scan order) if that
 * maximum is positive.  Same result as wesup_conv3x3_dgrad_winograd + wesup_maxpool2_bwd(accumulate = 1), which read and
 * re-wrote all four positions of every window. */
int wesup_conv3x3_dgrad_winograd_unpool(const float* dy, const float* u_dgrad, const float* unpool_src, float* unpool_dst,
                                        int B, int H, int W, int Hu, int Wu, int Cin, int Cout, int m,
                                        void* ws, size_t ws_bytes, void* stream);
/* The three passes on their own (the two entries above chain them): x (B,H,W,C) -> V [P][tiles][C];
 * nbatch NT products of one shape in one launch, C_b = A_b . B_b^T (element strides between the entries; K % 32 == 0);
 * Mt [P][tiles][C] -> y = A^T M A + bias, masked by mask_src > 0, added to the old y if accumulate, with the optional
 * second output y_relu = max(y, 0).  plane_elems: elements between two of the P position planes of V / Mt (0: tiles * C);
 * larger when a sub-batch works inside the planes of a whole batch, which is how the engine can pipeline the memory-bound
 * transforms of one half of the batch under the GEMM of the other. */
int wesup_winograd_input_transform(const float* x, float* V, long plane_elems, int B, int H, int W, int C, int relu_in,
                                   int m, void* stream);
int wesup_gemm_nt_batched(const float* A, int lda, long strideA, const float* B, int ldb, long strideB,
                          float* C, int ldc, long strideC, int nbatch, int M, int N, int K, void* stream);
/* ... with a bias vector per product (strideBias elements apart; bias may be NULL): several 1x1 side convs of one shape
 * (models/wesup.py:208-209,253: Conv2d(C, C/2, 1) on the hooked conv outputs of the layers that share a resolution) and their
 * input gradients as one launch each */
int wesup_gemm_nt_batched_bias(const float* A, int lda, long strideA, const float* B, int ldb, long strideB,
                               const float* bias, long strideBias, float* C, int ldc, long strideC,
                               int nbatch, int M, int N, int K, void* stream);
int wesup_winograd_output_transform(const float* Mt, long plane_elems, const float* bias, const float* mask_src, float* y,
                                    float* y_relu, float* y_pool, int pool_relu, int B, int H, int W, int C,
                                    int accumulate, int m, void* stream);
/* ... with the max-pool backward as its epilogue (the last pass of wesup_conv3x3_dgrad_winograd_unpool; m = 4) */
int wesup_winograd_output_transform_unpool(const float* Mt, long plane_elems, const float* bias, const float* mask_src,
                                           const float* unpool_src, float* unpool_dst, int B, int H, int W, int Hu, int Wu,
                                           int C, int m, void* stream);
/* The batched products and the output transform in ONE kernel (m = 4; K = 64 or 128 channels of the product, N % 64 == 0):
 * V [36][tiles][K] x U [36][N][K] -> y (B,H,W,N) = A^T (V_p . U_p^T) A + the epilogue of wesup_winograd_output_transform
 * (bias, mask_src, accumulate, y_pool) or of its _unpool form (unpool_src / unpool_dst (B,Hu,Wu,N), y = NULL): the
 * transformed output [36][tiles][N] is never written.  At these widths the separate passes are HBM-bound on exactly that
 * tensor; wesup_conv3x3_fwd/dgrad_winograd[_unpool] take this route by themselves (WESUP_WINO_FUSED=0: never, 1: forward only).
 * Results equal the two-kernel route's up to fp32 summation order. */
int wesup_winograd_fused_supported(int K, int N, int m);
/* ... and whether a problem of `tiles` tiles takes it (0: the grid of the one-kernel route would be too small, < 200 blocks) */
int wesup_winograd_fused_route(int K, int N, int m, long tiles);
int wesup_winograd_set_fused_min_blocks(int blocks);   /* tuning knob of that rule (default 200); returns the previous value */      /* 0: no; 1: the conv entries take the one-kernel route for this
                                                                 product in the forward; 2: in the input gradient as well */
int wesup_winograd_gemm_output_transform(const float* V, long plane_elems, const float* U, const float* bias,
                                         const float* mask_src, float* y, float* y_pool, int pool_relu,
                                         const float* unpool_src, float* unpool_dst, int Hu, int Wu,
                                         int B, int H, int W, int K, int N, int accumulate, void* stream);
/* ... with the destination's OLD content replaced by a gather: with the side conv of a native-resolution layer applied
 * behind the superpixel mean (the two commute, models/wesup.py:246-261,281-285), the side-branch gradient of that layer's
 * conv output is constant over a superpixel: side [B][Kmax][N] one row per superpixel, new_row [B][pixels] the pixel's row,
 * area_new [B][Kmax]; value(pixel) = side[new_row[pixel]] / area_new[new_row[pixel]] (what wesup_upsample_bwd would write).
 * y form: y = value(pixel) + (mask_src > 0 ? result : 0).  unpool form (unpool_src (B,Hu,Wu,N) given, y is the (B,Hu,Wu,N)
 * destination, Hu and Wu even): every position of a 2x2 window = its value, the first positive maximum of unpool_src gets
 * the window's result on top.  The destination is written, never read. */
int wesup_winograd_gemm_output_transform_gather(const float* V, long plane_elems, const float* U, const float* mask_src,
                                                float* y, const float* unpool_src, int Hu, int Wu, const float* side,
                                                const int32_t* new_row, const int32_t* area_new, int Kmax,
                                                int B, int H, int W, int K, int N, void* stream);
/* One pass over a gradient tensor for both of its F(4x4) transforms: V = B^T dY B (6x6 patches; the input of the layer's input
 * gradient) and dM = A dY A^T (4x4 cores; the second operand of its weight gradient), plus the per-block column sums of dy the
 * bias gradient is folded from (bias_part, optional: [wesup_winograd_bias_rows(B,H,W,C)][C]; 0 rows = this C is not covered).
 * wesup_conv3x3_wgrad_winograd_pre: the weight gradient from those operands and the forward's kept transformed input v_pre
 * (wesup_conv3x3_fwd_winograd's v_keep): batched TN products + filter-gradient reduce, the transform passes already done. */
long wesup_winograd_bias_rows(int B, int H, int W, int C);
int wesup_winograd_dual_transform(const float* dy, float* V, float* dM, float* bias_part, int B, int H, int W, int C,
                                  void* stream);
int wesup_conv3x3_wgrad_winograd_pre(const float* v_pre, const float* dm_pre, const float* bias_part, int bias_rows,
                                     float* dw_kcrs, float* db, int B, int H, int W, int Ci, int Cout,
                                     void* ws, size_t ws_bytes, void* stream);
/* Compact forms of what the backward needs of the forward's activations (F(4x4) route; 16x resp. 32x smaller than the tensors
 * they replace).  relu_bits [B][H][W][C/4] bytes: bit j of byte q = x[..., 4q + j] > 0 -- the ReLU decisions of the layer that
 * produced x, written by the input transform of the layer that consumes it (it reads x anyway).  pool codes
 * [B][H/2][W/2][C/4] uint16, 3 bits per channel of the quad: 0 = the window's maximum is not positive, k + 1 = first maximum at
 * window position k (row-major: torch's scan order) -- the max-pool's decisions, written by the epilogue that pools.
 * wesup_winograd_gemm_output_transform_ex: every option of the one-kernel route.  mask_bits instead of mask_src; pool_code
 * OUT (needs y_pool); unpool_code IN instead of reading unpool_src (then optional); side != NULL selects the gather forms. */
int wesup_winograd_input_transform_bits(const float* x, float* V, long plane_elems, unsigned char* relu_bits,
                                        int B, int H, int W, int C, int relu_in, void* stream);
int wesup_winograd_gemm_output_transform_ex(const float* V, long plane_elems, const float* U, const float* bias,
                                            const float* mask_src, const unsigned char* mask_bits, float* y, float* y_pool,
                                            int pool_relu, unsigned short* pool_code, const float* unpool_src,
                                            const unsigned short* unpool_code, float* unpool_dst, int Hu, int Wu,
                                            const float* side, const int32_t* new_row, const int32_t* area_new, int Kmax,
                                            int B, int H, int W, int K, int N, int accumulate, void* stream);
/* wesup_conv3x3_dgrad_winograd(accumulate = 1) resp. wesup_conv3x3_dgrad_winograd_unpool (unpool_src given; dx is then
 * (B,Hu,Wu,Cin)) with that gather in the epilogue instead of a materialised side-branch gradient in dx.  m = 4; product shapes
 * with wesup_winograd_fused_supported(Cout, Cin, 4) == 2, otherwise WESUP_ERR_INVALID (materialise, use the forms above).
 * area_new == NULL (here and in the two gather entries above): the rows of `side` are divided by their areas already --
 * wesup_scale_rows_by_area(side, area_new, B * Kmax, Cin): x[row][:] *= 1 / area[row] (0 where area is 0) -- so the epilogue
 * gathers one row per pixel and nothing else (conv1_2's input gradient alone 387 -> 349 us at configs[1]). */
int wesup_scale_rows_by_area(float* x, const int32_t* area, long rows, int C, void* stream);
int wesup_conv3x3_dgrad_winograd_gather(const float* dy, const float* u_dgrad, const float* mask_src,
                                        const float* unpool_src, float* dx, const float* side, const int32_t* new_row,
                                        const int32_t* area_new, int Kmax, int B, int H, int W, int Hu, int Wu,
                                        int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* The weight gradient's own transforms: dy (B,H,W,C) -> dM [P][tiles][C] = A dY A^T per m x m tile; and the way back from
 * the split-K slabs of the P transformed filter gradients ([P][S][Cout*Cin + Cout], each slab followed by the Cout
 * column sums of its dM operand) to dw (Cout,Cin,3,3) = G^T (sum over S) G.  The bias gradient db = sum over pixels of dy:
 * m = 2, the column sums of dM at position (1,1) (index 5; filter_grad's db); m = 4 (its point set 0, +-3/4, +-3/2, inf
 * has no position that is a plain sum), the optional db of the outgrad transform, summed from the values it loads anyway
 * (ws: wesup_winograd_outgrad_workspace_bytes; db must be NULL for m = 2 there, and for m = 4 in filter_grad). */
size_t wesup_winograd_outgrad_workspace_bytes(int B, int H, int W, int C, int m);
int wesup_winograd_outgrad_transform(const float* dy, float* dM, float* db, int B, int H, int W, int C, int m,
                                     void* ws, size_t ws_bytes, void* stream);
int wesup_winograd_filter_grad(const float* slabs, long slab_stride, long batch_stride, int S, float* dw_kcrs, float* db,
                               int Cout, int Cin, int m, void* stream);

/* ------------------------------------------------------------------ generic fp32 MFMA GEMMs
 * side 1x1 convs (models/wesup.py:208-209,253), fc_layers (models/wesup.py:213-220,288) and their grads.
 * nt:  C[M][N] = epilogue( A[M][K] . B[N][K]^T + bias[N] )      (K % 32 == 0, rows 16B aligned)
 * tn:  C[M][N] = A[K][M]^T . B[K][N]      (weight gradients; deterministic split-K)
 * colsum: out[N] = sum_m A[m][n]          (bias gradients without a weight gradient next to them) */
size_t wesup_gemm_nt_workspace_bytes(int M, int N, int K);      /* stream-K partial tiles; ws may be NULL */
int wesup_gemm_nt(const float* A, int lda, const float* B, int ldb, const float* bias,
                  float* C, int ldc, const float* mask, int ldmask,
                  int M, int N, int K, int flags, void* ws, size_t ws_bytes, void* stream);
size_t wesup_gemm_tn_workspace_bytes(int M, int N, int K);
/* colsum_a (optional, [M]): also out[m] = sum_k A[k][m] -- the bias gradient that goes with a weight gradient --
 * accumulated from the staged A tiles, so A is not read a second time.
 * Round 5: a product whose tiles fill the chip without splitting K (>= 600 tiles) runs ONE split and, without colsum_a (C 8-byte
 * aligned, ldc even), stores straight into C: no slab, no reduce launch; ws is still required (and may go unused). */
int wesup_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, float* colsum_a,
                  int M, int N, int K, int relu_b, void* ws, size_t ws_bytes, void* stream);
/* nbatch products of one shape in one launch: C_b = A_b^T . B_b, element strides between batch entries */
size_t wesup_gemm_tn_batched_workspace_bytes(int nbatch, int M, int N, int K);
int wesup_gemm_tn_batched(const float* A, int lda, long strideA, const float* B, int ldb, long strideB,
                          float* C, int ldc, long strideC, int nbatch, int M, int N, int K, int relu_b,
                          void* ws, size_t ws_bytes, void* stream);
size_t wesup_colsum_workspace_bytes(int M, int N);
int wesup_colsum(const float* A, int lda, float* out, int M, int N, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ pooling / upsampling (K2/K4)
 * maxpool: nn.MaxPool2d(2,2) after ReLU == ReLU after maxpool; operates on pre-ReLU y (relu_out applies the ReLU to
 * the pooled values as it stores them). */
int wesup_maxpool2_fwd(const float* y, float* yp, int B, int H, int W, int C, int relu_out /* yp = max(pool, 0) */,
                       void* stream);
/* dy[p] = (p is the first max of its window and y[p] > 0 ? dyp[window] : 0) (+ dy[p] if accumulate) */
int wesup_maxpool2_bwd(const float* y, const float* dyp, float* dy, int B, int H, int W, int C,
                       int accumulate, void* stream);
/* F.interpolate(bilinear, align_corners=True) (models/wesup.py:254-255) into channel slice
 * [coff, coff+C) of the feature-map tensor fm [B][H][W][ldf]. */
int wesup_upsample_fwd(const float* s, float* fm, int B, int h, int w, int H, int W, int C,
                       int ldf, int coff, void* stream);
/* ds[q][c] = sum_p wgt(p,q) * src(p)[coff+c].  src(p) = dfm[b][p][.] when new_row == NULL; otherwise the
 * pooling backward is fused in: src(p) = g[b][new_row[b][p]][.] / area[b][new_row[b][p]]  (ldf = row stride) */
int wesup_upsample_bwd(const float* dfm_or_g, const int32_t* new_row, const int32_t* area_new,
                       float* ds, int B, int h, int w, int H, int W, int C, int ldf, int coff,
                       int Kmax, void* stream);
/* The fused form (new_row given) for n <= 3 layers that share one coarse resolution (h, w) != (H, W), one launch:
 * ds_i [B][h][w][C_i] from g_i [B][Kmax][C_i] (dense rows, C_0 + .. + C_{n-1} <= 768); the scan of a cell's window of
 * full-resolution pixels is shared by the layers.  Same values as n calls of wesup_upsample_bwd. */
int wesup_upsample_bwd_group(const float* g0, const float* g1, const float* g2, float* ds0, float* ds1, float* ds2,
                             int C0, int C1, int C2, int n, const int32_t* new_row, const int32_t* area_new, int B,
                             int h, int w, int H, int W, int Kmax, void* stream);

/* ------------------------------------------------------------------ superpixels (K6/K8/K9)
 * wesup_sp_preprocess replaces _preprocess_superpixels (models/wesup.py:18-63) without dense maps.
 *   labels [B][HW] ids 0..n-1 (contiguous), mask [B][C][HW] uint8 {0,1} or NULL, Kmax >= max id+1
 * outputs (all [B][Kmax...] padded; rows are in the reference's order: labelled ids ascending, then unlabelled):
 *   n_sp[B], n_l[B], perm[B][Kmax] (row->id), inv_perm[B][Kmax] (id->row), area_new[B][Kmax] (by row),
 *   sp_labels[B][Kmax][C] f32 multi-hot (rows >= n_l zero), new_row[B][HW], row_start[B][Kmax+1],
 *   pix_sorted[B][HW] (pixels grouped by row, ascending pixel index inside a row),
 *   status[B]: 0 ok, 1 = label id >= Kmax, 2 = empty id below max (the reference would produce NaN). */
size_t wesup_sp_preprocess_workspace_bytes(int B, int HW, int C, int Kmax);
int wesup_sp_preprocess(const int32_t* labels, const uint8_t* mask, int B, int HW, int C, int Kmax,
                        int32_t* n_sp, int32_t* n_l, int32_t* perm, int32_t* inv_perm, int32_t* area_new,
                        float* sp_labels, int32_t* new_row, int32_t* row_start, int32_t* pix_sorted,
                        int32_t* status, int32_t* seg_start /* or NULL */, int32_t* unit_row /* or NULL */, int Umax,
                        void* ws, size_t ws_bytes, void* stream);
/* (the segment table of wesup_sp_segments is written on the way when seg_start / unit_row [B][Kmax+1] / [B][Umax] are given.
 *  Four launches, every phase parallel over chunks resp. ids: sums zeroed; per-chunk histograms in LDS, added to the image's
 *  sums with integer atomics (exact in any order); chunk prefixes beside the per-image ordering; placement.
 *  Limits: Kmax <= 16384 and Kmax * (2 + C) <= 81920 (LDS of the histogram pass: C <= 3 at Kmax = 16384), B <= 65535; no
 *  library-side state -- any number of streams may call it concurrently, each with a workspace of its own.) */
/* dense compat: labels[p] = argmax_n sp_maps[n][p] (first max), as models/wesup.py:295 does */
int wesup_spmaps_to_labels(const float* sp_maps, int32_t* labels, int N, int HW, void* stream);
/* scatter-mean: sp_feat[b][r][c] = (1/area_r) sum_{p in row r} fm[b][p][c]   (torch.mm, models/wesup.py:283-285) */
int wesup_sp_max_units(int HW, int Kmax);      /* upper bound of segments per image: Kmax + HW / 512 */
/* segment table for load-balanced pooling: a superpixel row is cut into segments of <= 512 pixels;
 * seg_start[B][Kmax+1] (exclusive scan of segments per row), unit_row[B][Umax] (segment -> row) */
int wesup_sp_segments(const int32_t* row_start, int B, int Kmax, int Umax, int32_t* seg_start, int32_t* unit_row,
                      void* stream);
size_t wesup_sp_pool_workspace_bytes(int B, int Umax, int C);   /* partial sums of multi-segment rows */
int wesup_sp_pool_fwd(const float* fm, const int32_t* pix_sorted, const int32_t* row_start,
                      const int32_t* seg_start, const int32_t* unit_row, float* sp_feat,
                      int B, int HW, int ldf, int C, int Kmax, int Umax, void* ws, size_t ws_bytes, void* stream);
/* fused upsample + scatter-mean: sp_feat[b][r][coff+c] = (1/area_r) sum_{p in row r} bilinear_ac(s[b], p)[c], s = side
 * output [B][h][w][C] (C in {32,64,128,256}); equals wesup_upsample_fwd followed by wesup_sp_pool_fwd on that slice
 * without materialising the (HW x 2112) feature map (models/wesup.py:254-261 + :283-285) */
int wesup_sp_pool_upsample_fwd(const float* s, const int32_t* pix_sorted, const int32_t* row_start,
                               const int32_t* seg_start, const int32_t* unit_row, float* sp_feat,
                               int B, int h, int w, int H, int W, int C, int ldo, int coff, int Kmax, int Umax,
                               void* ws, size_t ws_bytes, void* stream);

/* the same linear map as a matrix over the h*w cells of a coarse side output:
 * Wm[b][r][q] = (1/area_r) sum_{p in row r} bilinear_ac weight of cell q at pixel p     ([B][Kmax][h*w], h*w <= 8192).
 * Then  sp_feat[b][:, slice] = Wm[b] . s[b]  and  ds[b] = Wm[b]^T . g[b][:, slice]  are wesup_gemm_tn calls
 * (on Wm^T resp. Wm); used for the deep layers, where Wm is small (models/wesup.py:254-261 + :283-285 and autograd) */
int wesup_sp_interp_matrix(const int32_t* pix_sorted, const int32_t* row_start, float* Wm,
                           int B, int H, int W, int h, int w, int Kmax, void* stream);
/* dfm[b][p][c] = g[b][new_row[p]][c] / area[new_row[p]] */
int wesup_sp_pool_bwd(const float* g, const int32_t* new_row, const int32_t* area_new, float* dfm,
                      int B, int HW, int ldf, int C, int Kmax, void* stream);
/* paint-back (models/wesup.py:294-304): pred[b][p] = sp_pred[b][new_row[p]][cls] */
int wesup_paint_fwd(const float* sp_pred, const int32_t* new_row, float* pred, int B, int HW, int Kmax,
                    int C, int cls, void* stream);

/* ------------------------------------------------------------------ SLIC superpixels
 * replaces the CPU skimage.segmentation.slic call of WESUPTrainer.preprocess (models/wesup.py:471-476).
 * img [B][3][H][W] RGB in [0,1]; labels [B][HW] get contiguous ids 0..n_labels[b]-1 (4-connected superpixels,
 * numbered in raster order of their first pixel).  Third-party algorithm, parity unpinned (see csrc/slic.hip). */
int wesup_slic_num_centers(int H, int W, int n_segments);
size_t wesup_slic_workspace_bytes(int B, int H, int W, int n_segments);
int wesup_slic(const float* img_nchw, int32_t* labels, int32_t* n_labels, int B, int H, int W, int n_segments,
               float compactness, int max_iter, int enforce_connectivity, float min_size_factor,
               void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ head, loss, optimiser (K7/K10/K11/K13)
 * classifier Linear(D,2)+Softmax(dim=1) (models/wesup.py:229-232,292) */
int wesup_classifier_fwd(const float* feat, const float* Wc, const float* bc, float* pred, int R, int D, void* stream);
/* given dpred: dfeat[R][D] (masked by feat > 0: fc_layers' last ReLU), dWc[2][D], dbc[2] */
size_t wesup_classifier_bwd_workspace_bytes(int R, int D);
int wesup_classifier_bwd(const float* feat, const float* Wc, const float* pred, const float* dpred,
                         const float* dfeat_extra, float* dfeat, float* dWc, float* dbc, int R, int D,
                         void* ws, size_t ws_bytes, void* stream);
/* _label_propagate (models/wesup.py:99-139) per image on padded rows.  y_all[b][r][:] = sp_labels for
 * r < n_l, propagated pseudo label for n_l <= r < n_sp (zeros if max similarity <= threshold), 0 beyond.
 * src_idx/max_sim [B][Kmax] (entries for unlabelled rows; -1/0 elsewhere). */
int wesup_propagate(const float* feat, const float* sp_labels, const int32_t* n_sp, const int32_t* n_l,
                    float threshold, int enable, float* y_all, int32_t* src_idx, float* max_sim,
                    int B, int Kmax, int D, int C, void* stream);
/* _cross_entropy + compute_loss (models/wesup.py:66-96,492-531), per image then mean over images:
 * terms[b] = {sup_sum, sup_cnt, prop_sum, prop_cnt, prop_label_sum, loss_b, 0, 0}; loss[0] = mean_b loss_b = (sum over b in
 * ascending order, in fp32) / B -- or loss == NULL: no second launch, the caller forms that mean from terms */
int wesup_loss_fwd(const float* pred, const float* y_all, const int32_t* n_sp, const int32_t* n_l,
                   float eps, float prop_weight, float* terms, float* loss, int B, int Kmax, int C, void* stream);
/* dpred = dloss[0] * d loss / d pred */
int wesup_loss_bwd(const float* pred, const float* y_all, const int32_t* n_sp, const int32_t* n_l,
                   const float* terms, const float* dloss, float eps, float prop_weight, float* dpred,
                   int B, int Kmax, int C, void* stream);
/* ABI 5 -- the head of the training step in two launches instead of six (the chain between the fc layers' forward and their
 * backward is a chain of launch latencies).  Bit-identical to the entries they combine.
 * wesup_head_fwd  = wesup_classifier_fwd + wesup_propagate (both read feat (B*Kmax, D); pred (B*Kmax, 2)).
 * wesup_head_bwd  = wesup_loss_fwd (terms only) + wesup_loss_bwd + the first kernel of wesup_classifier_bwd: terms (B, 8),
 *                   dpred (B*Kmax, 2), dfeat (B*Kmax, D); Kmax % 64 == 0, C == 2; ws as wesup_classifier_bwd_workspace_bytes.
 * wesup_classifier_bwd_finish: dWc (2, D), dbc (2) from the partial sums wesup_head_bwd left in ws -- on any stream behind it
 *                   (models/wesup.py:66-139,229-232,492-531). */
int wesup_head_fwd(const float* feat, const float* Wc, const float* bc, float* pred, const float* sp_labels,
                   const int32_t* n_sp, const int32_t* n_l, float threshold, int enable, float* y_all, int32_t* src_idx,
                   float* max_sim, int B, int Kmax, int D, int C, void* stream);
int wesup_head_bwd(const float* feat, const float* Wc, const float* pred, const float* y_all, const int32_t* n_sp,
                   const int32_t* n_l, const float* dloss, float eps, float prop_weight, float* terms, float* dpred,
                   float* dfeat, int B, int Kmax, int D, int C, void* ws, size_t ws_bytes, void* stream);
int wesup_classifier_bwd_finish(const void* ws, size_t ws_bytes, float* dWc, float* dbc, int R, int D, void* stream);
/* generic _cross_entropy (models/wesup.py:66-96) on (n, C): out2[4] = {sum(-y log clamp(yhat) [* class_weights[c]]),
 * #rows with sum(y) > 0, loss = sum/#rows (0 if no row is labelled), 0}; class_weights (C,) or NULL (models/wesup.py:93-94) */
int wesup_cross_entropy_fwd(const float* y_hat, const float* y_true, const float* class_weights, float eps, float* out2,
                            int n, int C, void* stream);
int wesup_cross_entropy_bwd(const float* y_hat, const float* y_true, const float* class_weights, const float* out2,
                            const float* dloss, float eps, float* dy_hat, int n, int C, void* stream);
/* torch.optim.SGD step (models/wesup.py:445-451): g' = g*grad_scale + wd*p; v = first ? g' : mu*v + g'; p -= lr*v */
int wesup_sgd_step(float* p, const float* g, float* v, size_t n, float lr, float momentum, float weight_decay,
                   float grad_scale, int first_step, void* stream);
/* accuracy / dice inputs (utils/metrics.py:31-45,112-135): out[b] = {#(P==G), sum(P*G), sum(P), sum(G)} with
 * P = round(pred) (half to even, models/wesup.py:534), G = argmax_c mask (first max) */
size_t wesup_seg_metrics_workspace_bytes(int B);
int wesup_seg_metrics(const float* pred, const uint8_t* mask, float* out4, int B, int HW, int C,
                      void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ entries by the names of SURVEY.md 8(b)
 * One call per ATen op of the reference for a binding that replaces them one by one; each is a thin entry over the
 * kernels above (csrc/named.hip).  Matrices are row-major with the channel / feature index contiguous (NHWC pixels). */
/* K9 (models/wesup.py:34-42): area[B][Kmax] = pixels per superpixel id, counts[B][Kmax][C] = mask pixels per (id, class)
 * (counts may be NULL without a mask); status[b] |= 1 when an id is outside [0, Kmax) */
int wesup_sp_stats(const int32_t* labels, const uint8_t* mask, int B, int HW, int C, int Kmax, int32_t* area,
                   int32_t* counts, int32_t* status, void* stream);
/* K3 (models/wesup.py:208-209,253): Conv2d(Cin, Cout, 1) on P = B*h*w pixels; w [Cout][Cin]; w_t = w transposed */
size_t wesup_conv1x1_workspace_bytes(int P, int Cin, int Cout);
int wesup_conv1x1_fwd(const float* x, const float* w, const float* bias, float* y, int P, int Cin, int Cout,
                      void* ws, size_t ws_bytes, void* stream);
int wesup_conv1x1_dgrad(const float* dy, const float* w_t, float* dx, int P, int Cin, int Cout, int accumulate,
                        void* ws, size_t ws_bytes, void* stream);
int wesup_conv1x1_wgrad(const float* dy, const float* x, float* dw, float* db, int P, int Cin, int Cout,
                        void* ws, size_t ws_bytes, void* stream);
/* K7 (models/wesup.py:213-220,288): Linear(In, Out) [+ ReLU]; bwd: dw, db and (dx != NULL) dx = dy . w masked by
 * relu_src > 0 when the layer's input came out of a ReLU (relu_src = that input, [R][In]) */
size_t wesup_linear_workspace_bytes(int R, int In, int Out);
int wesup_linear_fwd(const float* x, const float* w, const float* bias, float* y, int R, int In, int Out, int relu,
                     void* ws, size_t ws_bytes, void* stream);
int wesup_linear_bwd(const float* dy, const float* x, const float* w_t, const float* relu_src, float* dx, float* dw,
                     float* db, int R, int In, int Out, void* ws, size_t ws_bytes, void* stream);
/* K4 (models/wesup.py:254-255): F.interpolate(mode='bilinear', align_corners=True), dense form, into / from the channel
 * slice [coff, coff + C) of a [B][H][W][ld_out] tensor */
int wesup_upsample_bilinear_ac_fwd(const float* s, float* out, int B, int h, int w, int H, int W, int C, int ld_out,
                                   int coff, void* stream);
int wesup_upsample_bilinear_ac_bwd(const float* dout, float* ds, int B, int h, int w, int H, int W, int C, int ld_out,
                                   int coff, void* stream);
/* K11 (models/wesup.py:231 Softmax + :66-96 _cross_entropy) fused: probs = softmax(logits) is written out, out2 as in
 * wesup_cross_entropy_fwd; bwd gives d loss / d logits */
int wesup_softmax_ce_fwd(const float* logits, const float* y_true, const float* class_weights, float eps, float* probs,
                         float* out2, int n, int C, void* stream);
int wesup_softmax_ce_bwd(const float* probs, const float* y_true, const float* class_weights, const float* out2,
                         const float* dloss, float eps, float* dlogits, int n, int C, void* stream);

/* ------------------------------------------------------------------ step plans (ABI 4)
 * The reference walks one training iteration in Python every step (models/base.py:184-211); so does the engine above this
 * library, ~330 launches on three streams.  A plan records that walk ONCE -- every launch this thread issues through the
 * library between wesup_plan_begin and wesup_plan_end is executed as usual and appended with its kernel, grid, stream and a
 * byte copy of its arguments; ordering edges and copies likewise -- and wesup_plan_replay re-issues the recording from C.
 * Nothing is re-derived at replay: the caller replays a plan only while every buffer the recorded step touched is alive at
 * the same address and the shapes / hyper-parameters are those of the recording (inputs go into buffers the plan knows). */
typedef struct WesupPlan WesupPlan;
int wesup_plan_create(WesupPlan** out /* host */);
int wesup_plan_destroy(WesupPlan* plan);
int wesup_plan_begin(WesupPlan* plan);            /* this thread records into plan (one at a time per thread) */
int wesup_plan_end(WesupPlan* plan);
int wesup_plan_size(const WesupPlan* plan);       /* nodes so far: a position to split a replay at */
int wesup_plan_kernels(const WesupPlan* plan);    /* kernel launches among them */
int wesup_plan_replay(const WesupPlan* plan, int first, int last);      /* nodes [first, last) in recorded order */
/* 0 = the two recordings are identical (kernels, geometry, streams, argument bytes, edges, copies); k > 0 = they first
 * differ at node k - 1: something the walk produces moved between the two steps and the older plan must not be replayed */
int wesup_plan_diff(const WesupPlan* a, const WesupPlan* b);
const char* wesup_plan_node_name(const WesupPlan* plan, int node);   /* kernel name, "<record s>", "<wait s>", "<copy n>" */
void* wesup_plan_node_stream(const WesupPlan* plan, int node);
/* diagnostics (environment WESUP_PLAN_TIMING=1): host nanoseconds the latest replay spent issuing the node, and when */
int wesup_plan_node_host_ns(const WesupPlan* plan, int node, long long* out2 /* host */);
/* Ordering edges between streams on a fixed pool of events addressed by slot (0 .. wesup_sync_slots() - 1): record marks
 * the work queued so far on `stream`, wait makes `stream` wait for the slot's latest mark.  Both act at once and, while the
 * thread records a plan, become nodes of it.  wesup_sync_synchronize blocks the HOST until the mark is reached (the one
 * wait of an iteration: the loss read-back). */
int wesup_sync_slots(void);
int wesup_sync_record(int slot, void* stream);
int wesup_sync_wait(int slot, void* stream);
int wesup_sync_synchronize(int slot);
int wesup_sync_query(int slot);                   /* 1 reached, 0 not yet */
/* recordable copies and fills: device -> pinned host, device -> device, 32-bit words */
int wesup_copy_to_host(void* dst_host_pinned, const void* src, size_t bytes, void* stream);
int wesup_copy(void* dst, const void* src, size_t bytes, void* stream);
int wesup_fill_words(void* ptr, uint32_t value, size_t words, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WESUP_HIP_H */
