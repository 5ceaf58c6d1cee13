/* y3d.h — C ABI of liby3d_hip.so: the MI355X (gfx950) kernels of the YOLOv10 / YOLOv10-3D hot path.
 *
 * The reference (baldhat/yolov10-3D) has no native layer: every entry point below replaces a
 * *torch op call site* inside one of the reference's nn.Modules / loss functions (file:line cited
 * per function, paths relative to ultralytics/).  The host-side mirror of those modules
 * (yolov10-3d_amd/modules.py, loss.py) binds this library through ctypes; INTEGRATION.md shows the
 * binding a reference maintainer would add.
 *
 * Conventions
 *   - all tensors are device pointers; activations are NHWC ("channels last"): element (b,h,w,c) of a
 *     tensor with strides (sb, sh, sw) lives at  ptr + b*sb + h*sh + w*sw + c   (strides in ELEMENTS).
 *     "pixel-dense" tensors only carry `sw` (pixel stride; sh = W*sw, sb = H*W*sw) so they may be
 *     channel slices of a wider buffer (concat-free writes).
 *   - dtype: Y3D_F32 (exact-f32 MFMA / fp32 VALU; the parity mode) or Y3D_BF16 (bf16 storage + bf16 MFMA,
 *     fp32 accumulation / statistics / loss math; the performance mode).
 *   - 16-byte chunks: channel counts, channel offsets and strides of activation tensors must be
 *     multiples of 4 (f32) / 8 (bf16) elements unless a function says otherwise.
 *   - `stream` is a hipStream_t (NULL = default stream).  Nothing here allocates, frees or synchronises:
 *     every call is capturable into a hipGraph.
 *   - return value: Y3D_OK or a negative Y3D_ERR_*; y3d_last_error() gives the message
 *     (the host mirror raises it as a Python exception — the reference's error behaviour).
 */
#ifndef Y3D_H
#define Y3D_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { Y3D_F32 = 0, Y3D_BF16 = 1 };
enum { Y3D_OK = 0, Y3D_ERR_INVALID = -1, Y3D_ERR_HIP = -2 };

const char* y3d_last_error(void);
int y3d_abi_version(void);
/* name of the device the library would launch on, its CU count and LDS bytes per workgroup (sanity / roofline) */
int y3d_device_info(char* name, int name_len, int* compute_units, int* lds_bytes, int* clock_khz);

/* ------------------------------------------------------------------------------------------------
 * Dense / grouped convolution, implicit GEMM on MFMA (conv_gemm.hip)
 * replaces nn.Conv2d.forward + autograd backward at nn/modules/conv.py:115,120-122 (Conv),
 * head.py:633-637 (3D head branches), block.py:225-226,336-337 (C2f / Bottleneck), ...
 * ---------------------------------------------------------------------------------------------- */
/* OIHW fp32 parameter -> K-contiguous compute-dtype layout [Cout][kh*kw][Cin_g_pad] */
int y3d_pack_weight_fwd(int dtype, const float* w_oihw, void* out, int Cout, int Cin_g, int Cin_g_pad, int kh, int kw, void* stream);
/* OIHW fp32 parameter -> [G][Cin_g][kh*kw][Cout_g] (row pitch y3d_conv_kpad(dtype, kh*kw*Cout_g)) for the data gradient */
int y3d_pack_weight_dgrad(int dtype, const float* w_oihw, void* out, int Cout, int Cin_g, int groups, int kh, int kw, void* stream);
/* the two packings above for MANY weights in one launch (once per step, after the optimizer): desc is a DEVICE array of 8 int64 per
 * weight {src fp32 OIHW, dst, a, b, c, taps, Kpad, mode}; mode 0 = forward layout (a = Cout, b = Cin/g, c = padded Cin/g),
 * mode 1 = data-gradient layout (a = groups, b = Cout/g, c = Cin/g); chunk tables as in the optimizer kernels below */
int y3d_mt_pack_weights(int dtype, const int64_t* desc, const int* chunk_tensor, const int* chunk_off, int nchunks, int chunk, void* stream);
/* fp8 conv weights (fp8w.hip; BASELINE configs[4], no reference counterpart): OCP e4m3fn codes with one power-of-two scale per
 * output channel, scale = 2^ceil(log2(absmax / 448)).  MANY weights in one launch: desc is a DEVICE array of 6 int64 per weight
 * {src fp32 (rows, K), w_eff fp32 (rows, K) or 0, codes uint8 (rows, K) or 0, scale fp32 (rows), rows, K}, row_begin (ntensors) the
 * first global row of each weight, nrows their total.  w_eff = value(code) * scale is exactly representable in bf16, so the bf16
 * matrix-core kernels compute fp8-weight products exactly when they pack w_eff instead of the master weight. */
int y3d_mt_fp8w_quantize(const int64_t* desc, const int* row_begin, int ntensors, int nrows, void* stream);
/* codes (rows, K) + scale (rows) -> fp32 weights (loading a 1-byte-per-weight checkpoint) */
int y3d_fp8w_dequantize(const uint8_t* codes, const float* scale, float* w, int rows, int K, void* stream);
/* fp8 MFMA convolution family (conv3x3_fp8.hip; BASELINE configs[4], no reference counterpart): 3x3 stride-1 "same" conv FORWARD on
 * v_mfma_scale_f32_16x16x128_f8f6f4 with OCP-MX operands.
 *   y3d_fp8_quantize_act: bf16 NHWC rows x (M pixels, row stride xsw elements, C % 32 == 0) -> e4m3 codes q (M, C) + E8M0 block scales s
 *     (M, y3d_fp8_scale_pitch(C)): C / 32 bytes per pixel, rows padded to whole dwords; scale = the smallest power of two with
 *     amax(32 channels) / scale <= 448, code = RNE(x / scale) (never saturates);
 *   y3d_fp8_pack_weight_fwd: the fp8w quantiser's codes (rows, Cg * 9) in OIHW order + power-of-two row scales -> K-contiguous bytes
 *     wq (rows, 9, Cg) + E8M0 bytes ws (rows);
 *   y3d_conv3x3_fp8_fwd: y (bf16 NHWC, pixel stride ysw) = conv(x, w) with fp32 accumulation; training form: stat_partials
 *     [y3d_conv3x3_fp8_stat_rows][Cout][2] (sum, sum^2 of y as stored), eval form: y = act(conv * scale[c] + shift[c]).
 *     Served geometries: y3d_conv3x3_fp8_ok (Cin / groups a multiple of 64, at least 128; Cout / groups a multiple of 16). */
int y3d_fp8_quantize_act(const void* x, int64_t xsw, int64_t M, int C, uint8_t* q, uint8_t* s, void* stream);
int y3d_fp8_pack_weight_fwd(const uint8_t* codes, const float* scale, int rows, int Cg, uint8_t* wq, uint8_t* ws, void* stream);
int y3d_conv3x3_fp8_ok(int B, int H, int W, int Cin, int Cout, int groups);
int y3d_conv3x3_fp8_stat_rows(int B, int H, int W);
int y3d_fp8_scale_pitch(int C);
int y3d_conv3x3_fp8_fwd(const uint8_t* xq, const uint8_t* xs, int B, int H, int W, int Cin, const uint8_t* wq, const uint8_t* ws, void* y, int64_t ysw,
                        int Cout, int groups, float* stat_partials, const float* scale, const float* shift, int act, void* stream);
int y3d_conv_kpad(int dtype, int k_total);
/* number of BatchNorm partial rows of the generic implicit-GEMM kernel: ceil(B*Ho*Wo / 128) */
int y3d_conv_stat_blocks(int B, int Ho, int Wo);
/* number of BatchNorm partial rows y3d_conv2d_fwd emits for this geometry (depends on the kernel it dispatches to:
 * 3x3 s1 p1 convs with 128-byte channel slabs run the resident-halo tile kernel, one row per TH x 16 pixel tile) */
int y3d_conv2d_stat_rows(int dtype, int B, int H, int W, int Cin, int Cout, int groups, int kh, int kw, int stride, int pad);
/* y = conv(x, w) (+bias).  stat_partials (optional, no bias): [y3d_conv_stat_blocks][Cout][2] = per-block (sum, sum^2) of y. */
int y3d_conv2d_fwd(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int Cin,
                   const void* w_packed, const float* bias, void* y, int64_t ysw, int Ho, int Wo, int Cout, int groups,
                   int kh, int kw, int stride, int pad, float* stat_partials, void* stream);
/* eval-mode Conv in ONE launch: y = act(conv(x, w) * scale[c] + shift[c]) with scale/shift from y3d_bn_eval_scale
 * (Conv.forward with BatchNorm in eval mode, conv.py:120-122; equals the reference's fuse_conv_and_bn folding, torch_utils.py:171-198) */
int y3d_conv2d_fwd_affine(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int Cin,
                          const void* w_packed, const float* scale, const float* shift, int act, void* y, int64_t ysw, int Ho, int Wo,
                          int Cout, int groups, int kh, int kw, int stride, int pad, void* stream);
/* the same with the residual of a shortcut block added AFTER the activation: y = act(conv(x, w) * scale + shift) + res
 * (Bottleneck.forward in eval mode, block.py:342).  Only geometries for which y3d_conv2d_fwd_affine_res_ok returns 1 (bf16 3x3 s1 p1,
 * <= 64 input and output channels: the narrow resident-weight kernel); other callers run y3d_conv2d_fwd + y3d_bn_act_fwd. */
int y3d_conv2d_fwd_affine_res_ok(int dtype, int B, int H, int W, int Cin, int Cout, int groups, int kh, int kw, int stride, int pad);
int y3d_conv2d_fwd_affine_res(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int Cin,
                              const void* w_packed, const float* scale, const float* shift, int act, const void* res, int64_t rsw, void* y,
                              int64_t ysw, int Ho, int Wo, int Cout, int groups, int kh, int kw, int stride, int pad, void* stream);
/* dx = conv_transpose(dy, w)  (F.conv2d backward w.r.t. input) */
int y3d_conv2d_bwd_data(int dtype, const void* dy, int64_t dsb, int64_t dsh, int64_t dsw, int B, int Ho, int Wo, int Cout,
                        const void* w_packed_dgrad, void* dx, int64_t xsw, int H, int W, int Cin, int groups, int kh, int kw,
                        int stride, int pad, void* stream);
/* split-K factor of the generic weight-gradient kernel; slab must hold nsplit*Cout*kh*kw*Cin_g floats */
int y3d_conv2d_wgrad_splits(int dtype, int B, int Ho, int Wo, int Cout, int Cin_g, int groups, int kh, int kw);
/* split-K factor y3d_conv2d_bwd_weight expects for this geometry (it dispatches 3x3 s1 p1 bf16 convs with 64-channel slabs to
 * the resident-tile weight-gradient kernel); same slab sizing rule */
int y3d_conv2d_wgrad_plan(int dtype, int B, int H, int W, int Cin, int Cout, int groups, int kh, int kw, int stride, int pad);
/* testing / A-B knob: 0 routes every convolution through the generic implicit-GEMM kernels; returns the previous value */
int y3d_set_tile_kernels(int enable);
/* same for the streaming 1x1 kernels (conv1x1_stream.hip: bf16 1x1 stride-1 forward / data gradient, LDS-resident weights;
 * wgrad1x1_stream.hip: its weight gradient, LDS-DMA ring) */
int y3d_set_stream1x1(int enable);
int y3d_get_stream1x1(void);
int y3d_get_tile_kernels(void);
/* same for the row-wide workgroup slabs of the BatchNorm forward / backward-apply passes on bf16 tensors with >= 512 channels
 * (bn_act.hip: slab_width) */
int y3d_set_bn_wide_slabs(int enable);
/* grad_oihw (+)= dL/dw.  Cin may be channel-padded (stem): only the first Cin_real channels are written. */
int y3d_conv2d_bwd_weight(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int Cin,
                          int Cin_real, const void* dy, int64_t dsw, int Ho, int Wo, int Cout, int groups, int kh, int kw,
                          int stride, int pad, float* slab, int nsplit, float* grad_oihw, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Depth-wise convolution (dwconv.hip) — Conv(g=c): conv.py:172-177, block.py:705-706,747-751,783,824
 * ---------------------------------------------------------------------------------------------- */
int y3d_dw_blocks(int64_t M);        /* rows of BN partials for M output pixels */
int y3d_dw_wgrad_blocks(int64_t M);  /* weight-gradient slabs for M output pixels */
int y3d_dw_pack_weight(const float* w_oihw, float* out_taps_c, int C, int kh, int kw, void* stream);
int y3d_dwconv2d_fwd(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int C,
                     const float* w_packed, void* y, int64_t ysw, int Ho, int Wo, int kh, int kw, int stride, int pad,
                     float* stat_partials, void* stream);
/* eval: z = act(dwconv(x) * scale[c] + shift[c] (+ res, res_mode 2)) (+ res, res_mode 1) in one launch - the folded BatchNorm + SiLU (+ the
 * RepVGGDW / shortcut residual, block.py:711, 758) applied to the depth-wise result rounded as y3d_dwconv2d_fwd would have stored it, so the
 * values are those of y3d_dwconv2d_fwd + y3d_bn_act_fwd bit for bit */
int y3d_dwconv2d_fwd_affine(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int C,
                            const float* w_packed, const float* scale, const float* shift, int act, int res_mode, const void* res, int64_t rsw,
                            void* z, int64_t zsw, int Ho, int Wo, int kh, int kw, int stride, int pad, void* stream);
int y3d_dwconv2d_bwd_data(int dtype, const void* dy, int64_t dsb, int64_t dsh, int64_t dsw, int B, int Ho, int Wo, int C,
                          const float* w_packed, void* dx, int64_t xsw, int H, int W, int kh, int kw, int stride, int pad,
                          void* stream);
/* slab: y3d_dw_wgrad_blocks(B*Ho*Wo) * kh*kw*C floats */
int y3d_dwconv2d_bwd_weight(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int C,
                            const void* dy, int64_t dsw, int Ho, int Wo, int kh, int kw, int stride, int pad, float* slab,
                            float* grad_oihw, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm (batch or running statistics) + SiLU + residual (bn_act.hip)
 * replaces nn.BatchNorm2d / nn.SiLU / `x + ...` at conv.py:106,118-122, block.py:342,711,758,816-817;
 * eps / momentum semantics of utils/torch_utils.py:327-337
 * ---------------------------------------------------------------------------------------------- */
/* partials [nblk][C][2] -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale; running stats
 * updated in place (momentum, unbiased variance) when running_mean != NULL */
int y3d_bn_finalize(const float* partials, int nblk, int C, int64_t count, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                    float* shift, void* stream);
int y3d_bn_eval_scale(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, float* scale, float* shift, void* stream);
/* u = y*scale+shift (+res if res_mode==2); z = act ? silu(u) : u; (+res if res_mode==1) */
int y3d_bn_act_fwd(int dtype, const void* y, int64_t ysw, const float* scale, const float* shift, int act, int res_mode,
                   const void* res, int64_t rsw, void* z, int64_t zsw, int64_t P, int C, void* stream);
/* y3d_bn_act_fwd (bf16, no residual, C % 64 == 0) that ALSO writes the fp8 copy of z the next layer's fp8 MFMA convolution reads:
 * q (P, C) e4m3 codes + s (P, y3d_fp8_scale_pitch(C)) E8M0 block scales, exactly what y3d_fp8_quantize_act makes of z */
int y3d_bn_act_fwd_q(const void* y, int64_t ysw, const float* scale, const float* shift, int act, void* z, int64_t zsw, uint8_t* q, uint8_t* s,
                     int64_t P, int C, void* stream);
int y3d_bn_bwd_blocks(int64_t P, int C);
/* pass 1: partials [y3d_bn_bwd_blocks(P, C)][C][2] = (sum g, sum g*xhat), g = dz * act'(u) */
int y3d_bn_act_bwd_reduce(int dtype, const void* y, int64_t ysw, const void* dz, int64_t dsw, const void* res, int64_t rsw,
                          const float* scale, const float* shift, const float* mean, const float* invstd, int act,
                          int res_mode, float* partials, int64_t P, int C, void* stream);
/* dgamma/dbeta (+)= sums; mean_g, mean_gx = sums / count */
int y3d_bn_bwd_finalize(const float* partials, int nblk, int C, int64_t count, float* dgamma, float* dbeta, int accumulate,
                        float* mean_g, float* mean_gx, void* stream);
/* pass 2: dy = scale*(g - mean_g - xhat*mean_gx) [train] or scale*g [eval]; dres = g for res_mode==2 */
int y3d_bn_act_bwd_apply(int dtype, const void* y, int64_t ysw, const void* dz, int64_t dsw, const void* res, int64_t rsw,
                         const float* scale, const float* shift, const float* mean, const float* invstd,
                         const float* mean_g, const float* mean_gx, int act, int res_mode, int train, void* dy, int64_t dysw,
                         void* dres, int64_t drsw, int64_t P, int C, void* stream);
/* column sums of a [P][C] tensor as partials [y3d_bn_bwd_blocks(P, C)][C][2] (bias gradients; reduce with y3d_bn_bwd_finalize) */
int y3d_colsum_partials(int dtype, const void* x, int64_t xsw, float* partials, int64_t P, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Graph glue (misc.hip)
 * ---------------------------------------------------------------------------------------------- */
/* nn.MaxPool2d(k, 1, k//2) — SPPF block.py:171-178.  argmax: [B*H*W][C] uint8 tap index (may be NULL) */
int y3d_maxpool_fwd(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, void* y, int64_t ysw, uint8_t* argmax,
                    int B, int H, int W, int C, int k, void* stream);
int y3d_maxpool_bwd(int dtype, const void* dy, int64_t dsw, const uint8_t* argmax, void* dx, int64_t xsw, int B, int H, int W,
                    int C, int k, void* stream);
/* nn.Upsample(None, 2, "nearest") — cfg/models/v10 and v10-3D yaml rows; x is (B,H,W,C), y is (B,2H,2W,C) */
int y3d_upsample2x_fwd(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, void* y, int64_t ysw, int B, int H, int W,
                       int C, void* stream);
int y3d_upsample2x_bwd(int dtype, const void* dy, int64_t dsb, int64_t dsh, int64_t dsw, void* dx, int64_t xsw, int B, int H,
                       int W, int C, void* stream);
/* torch.cat(dim=1) member copy — Concat conv.py:404, block.py:178,233,818: y[:, :C] = x over P pixels */
int y3d_copy2d(int dtype, const void* x, int64_t xsw, void* y, int64_t ysw, int64_t P, int C, void* stream);
/* diagnostic, no reference counterpart (tools/cu_contention.py): n workgroups that stay resident on `stream` until *flag != 0 (device
 * int) or ~50 ms, moving a trickle of data inside buf - a stand-in for an RCCL kernel's CU footprint next to the backward pass */
int y3d_occupy_cus(int n, const int* flag, float* buf, int64_t nbuf, void* stream);
int y3d_add2d(int dtype, const void* a, int64_t asw, const void* b, int64_t bsw, void* y, int64_t ysw, int64_t P, int C, void* stream);
/* model boundary: reference tensors are NCHW fp32 (nn/tasks.py:93-95) */
int y3d_nchw_to_nhwc(int dtype, const float* x_nchw, void* y_nhwc, int B, int C, int H, int W, int Cpad, void* stream);
/* stem (nn/tasks.py yaml row 0: Conv(3, c, 3, 2)): NCHW fp32 image -> [B][Ho][Wo][32], column (r*3+q)*3+ci of the 3x3 stride-2 pad-1
 * window, columns 27..31 zero; the stem conv is then y3d_conv2d_fwd / _bwd_weight with a 1x1 kernel over 32 channels */
int y3d_stem_im2col(int dtype, const float* x_nchw, void* out, int B, int H, int W, int Ho, int Wo, void* stream);
/* the same from the dataset's uint8 image (NCHW: hwc = 0, NHWC as decoded: hwc = 1) with the /255 of data/datasets/kitti.py:204-205
 * (models/yolo/detect/train.py:59 for the 2D trainer) done on the device: 1 byte per sample crosses PCIe / HBM instead of 4 */
int y3d_stem_im2col_u8(int dtype, const uint8_t* x, int hwc, void* out, int B, int H, int W, int Ho, int Wo, void* stream);
/* the eval stem in one pass - Conv(3, Cout, 3, 2) with running-statistics BatchNorm folded into (scale, shift) and SiLU (act != 0),
 * nn/modules/conv.py:120-122 on yaml row 0: x = (B, 3, H, W) fp32 (in_mode 0), (B, 3, H, W) uint8 (1) or (B, H, W, 3) uint8 (2; uint8
 * samples are divided by 255 as data/datasets/kitti.py:204-205); wcol = [Cout][32] fp32 with column (r*3+q)*3+ci of the 3x3 window
 * (27..31 zero); y = bf16 (B, Ho, Wo) pixels x Cout channels (16, 32, 48, 64 or 80), pixel pitch ysw elements.  bf16 arithmetic as
 * y3d_stem_im2col + y3d_conv2d_fwd_affine, without the column tensor. */
int y3d_stem_conv_eval(const void* x, int in_mode, const float* wcol, const float* scale, const float* shift, int act, void* y, int64_t ysw,
                       int B, int H, int W, int Cout, void* stream);
/* the training stem in one pass: the same gather + MFMA writing the raw conv output y (bf16, before BatchNorm), the column tensor
 * xcol = [B*Ho*Wo][32] bf16 that y3d_stem_im2col would have produced (operand of the weight gradient) and BatchNorm partials
 * part[rows][Cout][2] (sum, sum of squares of the stored values; rows = y3d_stem_conv_train_rows) for y3d_bn_finalize -
 * replaces y3d_stem_im2col + y3d_conv2d_fwd (K = 32), which wrote the column tensor and read it back */
int y3d_stem_conv_train_rows(int B, int H, int W);
int y3d_stem_conv_train(const void* x, int in_mode, const float* wcol, void* y, int64_t ysw, void* xcol, float* part, int B, int H, int W,
                        int Cout, void* stream);
int y3d_nhwc_to_nchw(int dtype, const void* x_nhwc, int64_t xsw, float* y_nchw, int B, int C, int H, int W, void* stream);
/* head final nn.Conv2d(c, out, 1) with bias, out <= 24 — head.py:637 (3D branches: nc,2,2,2,3,24,1,1) */
int y3d_proj_fwd(int dtype, const void* x, int64_t xsw, const float* w, const float* bias, void* y, int64_t ysw, int64_t P,
                 int Cin, int Cout, void* stream);
int y3d_proj_bwd_data(int dtype, const void* dy, int64_t dsw, const float* w, void* dx, int64_t xsw, int64_t P, int Cin, int Cout,
                      void* stream);
int y3d_proj_blocks(int64_t P);
/* slab: y3d_proj_blocks(P)*Cout*Cin floats, bias_slab: y3d_proj_blocks(P)*Cout floats */
int y3d_proj_bwd_weight(int dtype, const void* x, int64_t xsw, const void* dy, int64_t dsw, float* slab, float* bias_slab,
                        float* grad_w, float* grad_b, int accumulate, int64_t P, int Cin, int Cout, void* stream);
int y3d_slab_reduce(const float* slab, float* out, int nblk, int64_t n, int accumulate, void* stream);
/* the nb (<= 16) final projections of one head level in ONE launch (proj_group.hip): branch j reads channels
 * [xoff[j], xoff[j]+cin) of x and writes couts[j] channels at the running output offset (the torch.cat of head.py:742).
 * w / b / dw / db: host arrays of nb device pointers.  slab: y3d_proj_group_blocks(P)*sum(couts)*cin floats, bslab: ...*sum(couts) */
int y3d_proj_group_fwd(int dtype, int nb, int cin, const void* x, int64_t xsw, const int* xoff, const float* const* w,
                       const float* const* b, const int* couts, void* y, int64_t ysw, int64_t P, void* stream);
int y3d_proj_group_bwd_data(int dtype, int nb, int cin, const void* dy, int64_t dsw, const int* xoff, const float* const* w,
                            const int* couts, void* dx, int64_t xsw, int64_t P, void* stream);
/* BatchNorm backward of the conv that feeds a projection group, recomputing dz = dout . W on the matrix cores instead of reading a
 * materialised gradient tensor (proj_bn_mfma.hip, bf16, cin = 64 / 128; the backward of head.py:633-637 + conv.py:120 for one level):
 * mode 0: partials [y3d_proj_group_bn_bwd_blocks(P)][C][2] = (sum g, sum g*xhat), g = dz * act'(u)  -> y3d_bn_bwd_finalize;
 * mode 1: dy = scale * (g - mean_g - xhat * mean_gx).  y_pre: the conv's pre-BatchNorm output (C channels, branch i at xoff[i]),
 * dout: gradient of the projected map (branch i's couts[i] channels side by side in branch order). */
/* the forward of the same stage on the matrix cores (bf16, cin = 64 / 128): out[px][branch i's channels] = W_i . act(y_pre * scale + shift) + b_i */
int y3d_proj_group_fwd_bn_mfma(int nb, int cin, const void* y_pre, int64_t ysw, const int* xoff, const float* const* w, const float* const* b,
                               const int* couts, const float* scale, const float* shift, int act, void* out, int64_t osw, int64_t P,
                               void* stream);
/* ... and its weight / bias gradients (contraction over pixels; transposed LDS reads): slab = blocks * sum(couts) * cin floats,
 * bslab = blocks * sum(couts) floats, blocks = y3d_proj_group_bwd_weight_bn_mfma_blocks(P) */
int y3d_proj_group_bwd_weight_bn_mfma_blocks(int64_t P);
int y3d_proj_group_bwd_weight_bn_mfma(int nb, int cin, const void* y_pre, int64_t ysw, const int* xoff, const void* dout, int64_t dsw,
                                      const int* couts, const float* scale, const float* shift, int act, float* slab, float* bslab,
                                      float* const* dw, float* const* db, int64_t P, void* stream);
int y3d_proj_group_bn_bwd_blocks(int64_t P);
int y3d_proj_group_bn_bwd(int mode, int nb, int cin, const void* y_pre, int64_t ysw, const int* xoff, const void* dout, int64_t dsw,
                          const float* const* w, const int* couts, const float* scale, const float* shift, const float* mean,
                          const float* invstd, const float* mean_g, const float* mean_gx, int act, float* partials, int nblk, void* dy,
                          int64_t dysw, int64_t P, int C, void* stream);
int y3d_proj_group_blocks(int64_t P);
/* the same projections fused with the BatchNorm (+SiLU) that precedes them: y_pre is the PRE-BatchNorm conv output; the activation
 * act(y_pre * scale + shift) is formed on the fly (forward) / rebuilt (weight gradient) and never stored */
int y3d_proj_group_fwd_bn(int dtype, int nb, int cin, const void* y_pre, int64_t xsw, const int* xoff, const float* const* w,
                          const float* const* b, const int* couts, const float* scale, const float* shift, int act, void* out,
                          int64_t ysw, int64_t P, void* stream);
int y3d_proj_group_bwd_weight_bn(int dtype, int nb, int cin, const void* y_pre, int64_t xsw, const int* xoff, const void* dy, int64_t dsw,
                                 const int* couts, const float* scale, const float* shift, int act, float* slab, float* bslab,
                                 float* const* dw, float* const* db, int64_t P, void* stream);
int y3d_proj_group_bwd_weight(int dtype, int nb, int cin, const void* x, int64_t xsw, const int* xoff, const void* dy, int64_t dsw,
                              const int* couts, float* slab, float* bslab, float* const* dw, float* const* db, int64_t P, void* stream);

/* ------------------------------------------------------------------------------------------------
 * PSA attention core (attn.hip) — Attention.forward block.py:785-797
 * qkv: (B, N, nh*(2kd+hd)) with per-head block [q|k|v]; out: (B, N, nh*hd); lse/delta: (B, nh, N) fp32
 * ---------------------------------------------------------------------------------------------- */
int y3d_attn_fwd(int dtype, const void* qkv, int64_t qsw, void* out, int64_t osw, float* lse, int B, int N, int nh, int kd, int hd,
                 float scale, void* stream);
int y3d_attn_bwd(int dtype, const void* qkv, int64_t qsw, const void* out, int64_t osw, const void* dout, int64_t dsw,
                 const void* dv_extra, int64_t esw, const float* lse, float* delta, void* dqkv, int64_t gsw, int B, int N, int nh,
                 int kd, int hd, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 3D task-aligned assignment + 3D detection loss (tal_loss3d.hip)
 * replaces TaskAlignedAssigner3d.forward utils/tal.py:392-452 (+ keypoint_utils.py:11-118, metrics.py:78-134) and
 * DDDetectionLoss.__call__ utils/loss.py:821-963 for one head set.
 * maps[l]: (B, H[l], W[l], >= nc+35) NHWC head map of level l (pointer at channel 0 of the head set, pixel stride psw[l]);
 * anchors are level-major, row-major (tal.py:300-312).  gt: (B, n, 17) padded targets of loss.py:795-810
 * (cls | box xyxy px | center_2d | size_2d | center_3d | size_3d | depth | heading_bin | heading_res).
 * ---------------------------------------------------------------------------------------------- */
int y3d_tal3d_scratch_floats(int B, int n, int A, int topk);
/* `preprocess` utils/loss.py:795-810 (3D: :848-856, 2D: :223-226): rows (nbox, 1+width) = [batch_idx | cls | box xywh in [0,1] | ...] ->
 * out (B, cap, width) zero-padded per image in order of appearance, box scaled by (scale_x, scale_y) and converted to xyxy px.
 * n_used: TWO device ints.  n_used[0] = min(largest per-image box count, cap): the assigners take it as a DEVICE pointer, so —
 * unlike the reference's host-side `counts.max()` — padding the targets needs no host synchronisation.  n_used[1] = the largest
 * count itself: boxes beyond `cap` per image are dropped, and a caller must treat n_used[1] > cap as an error (loss.pad_targets
 * reads it back asynchronously and raises). */
int y3d_pad_targets(const float* rows, int nbox, int width, int B, int cap, float scale_x, float scale_y, float* out, int* n_used,
                    void* stream);
/* outputs: fg_mask (B,A) uint8, target_gt_idx (B,A) int32, target_scores (B,A,nc) fp32 (normalised),
 * scal[0] = max(sum(target_scores), 1), scal[1] = number of foreground anchors.
 * n = rows per image of gt (capacity, <= 64); n_used: device int from y3d_pad_targets (rows >= *n_used are padding in every
 * image and are skipped), or NULL = all n rows are walked.  The results do not depend on n_used. */
/* mode (cfg/default.yaml:116-119 -> utils/tal.py:465-497): bit 0 `tal_2d` (box metric), bit 1 `tal_3d` (keypoint metric; at least one of
 * the two), bit 2 `kps_dist_metric: l2` (else l1), bit 3 `constrain_anchors` (candidates inside the box only).  Default 1 | 2 | 8 = 11. */
int y3d_tal3d_assign(int dtype, int nl, const void* const* maps, const int64_t* psw, const int* H, const int* W, const float* strides,
                     int B, int nc, const float* gt, int n, const float* calib, const float* mean_sizes, int topk, float alpha,
                     float beta, float gamma, int mode, float* scratch, uint8_t* fg_mask, int* target_gt_idx, float* target_scores, float* scal,
                     const int* n_used, void* stream);
/* items[6] = (box2d, cls, depth, offset3d, size3d, heading) of loss.py:886-891; grads[l] (pixel stride gsw[l]) receives
 * grad_scale * d(sum(items))/d(map) for all nc+35 channels.  partials: 6 * ceil(B*A/256) floats */
int y3d_loss3d(int dtype, int nl, const void* const* maps, const int64_t* psw, void* const* grads, const int64_t* gsw, const int* H,
               const int* W, const float* strides, int B, int nc, const float* gt, int n, const uint8_t* fg_mask,
               const int* target_gt_idx, const float* target_scores, const float* scal, float w_loss2d, float w_cls, float w_depth,
               float w_offset3d, float w_size3d, float w_heading, float grad_scale, float* partials, float* items, void* stream);

/* 2D counterparts (tal_loss2d.hip): TaskAlignedAssigner utils/tal.py:45-94 and v8DetectionLoss utils/loss.py:206-257 (+BboxLoss :82-113,
 * DFL block.py:59-62).  maps[l]: (B, H, W, 64 + nc) with channel order [4 x 16 DFL bins | nc class logits]; gt: (B, n, 5) = cls | box xyxy px.
 * scratch as y3d_tal3d_scratch_floats.  items[3] = (box, cls, dfl) incl. gains; partials: 3 * ceil(B*A/256) floats */
int y3d_tal2d_assign(int dtype, int nl, const void* const* maps, const int64_t* psw, const int* H, const int* W, const float* strides,
                     int B, int nc, const float* gt, int n, int topk, float alpha, float beta, float* scratch, uint8_t* fg_mask,
                     int* target_gt_idx, float* target_scores, float* scal, const int* n_used, void* stream);
int y3d_loss2d(int dtype, int nl, const void* const* maps, const int64_t* psw, void* const* grads, const int64_t* gsw, const int* H,
               const int* W, const float* strides, int B, int nc, const float* gt, int n, const uint8_t* fg_mask,
               const int* target_gt_idx, const float* target_scores, const float* scal, float w_box, float w_cls, float w_dfl,
               float grad_scale, float* partials, float* items, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Eval-side selection (post.hip): v10Detect3d.select_candidates / extract_patches / scatter / decode (head.py:656-716,
 * 755-797) and v10_3Dpostprocess / v10postprocess (utils/ops.py:852-880).  Ties go to the lowest index.
 * ---------------------------------------------------------------------------------------------- */
/* out_idx (B, K) int32: flat cell index (row*W + col) of the K largest max-class logits of each image; cls: (B, HW, >= nc), pixel stride psw */
int y3d_topk_cells(int dtype, const void* cls, int64_t psw, int B, int HW, int nc, int K, int* out_idx, void* stream);
/* out (B*K, ps, ps, C) NHWC: zero-padded ps x ps input patches centred on the selected cells */
int y3d_patch_gather(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, const int* idx, void* out, int B, int H, int W,
                     int C, int K, int ps, void* stream);
/* map (B, HW, no): dense cls channels + the (no - nc) regression channels of the candidate on each selected cell, zeros elsewhere */
int y3d_head3d_scatter(int dtype, const void* cls, int64_t csw, const void* reg, int64_t rsw, const int* idx, void* map, int B, int HW,
                       int nc, int no, int K, void* stream);
/* y (B, nc+35, A) fp32 from the per-level (B, H, W, nc+35) maps: xyxy px boxes, centre-3d px, the rest copied */
int y3d_head3d_decode(int dtype, int nl, const void* const* maps, const int* H, const int* W, const float* strides, int B, int nc,
                      float* y, void* stream);
/* 2D head decode — Detect.inference head.py:53-79 (DFL softmax expectation block.py:59-62, dist2bbox(xywh) tal.py:315-325, sigmoid):
 * maps[l] (B, H, W, 64 + nc) NHWC pixel-dense with channels [4 x 16 box-side bins | nc class logits] -> y (B, 4 + nc, A) fp32 =
 * (cx, cy, w, h) px | scores */
int y3d_head2d_decode(int dtype, int nl, const void* const* maps, const int* H, const int* W, const float* strides, int B, int nc,
                      float* y, void* stream);
/* y (B, C, A) fp32; scores are the first nc rows (3D, boxes_first = 0) or the last nc rows (2D, boxes_first = 1).
 * reg (B, max_det, C-nc), scores (B, max_det), labels (B, max_det) int64 */
int y3d_v10_postprocess_scratch_floats(int B, int A, int nc, int max_det); /* 0 when the score row fits LDS, else B*A */
int y3d_v10_postprocess(const float* y, int B, int C, int A, int nc, int max_det, int boxes_first, float* reg, float* scores,
                        int64_t* labels, float* scratch, void* stream);

/* Input pipeline, image side of KITTIDataset.__getitem__ (data/datasets/kitti.py:132-206): mirror (:149, :184), mixup blend
 * `Image.blend(img, img2, 0.5)` (:188), affine crop `img.transform(resolution, AFFINE, trans_inv, BILINEAR)` (:192-196) in
 * Pillow's arithmetic (bit-exact: double-precision mapping of pixel centres, clamped bilinear taps, truncation), `/255` (:204).
 * src / src2: DEVICE arrays of B device pointers to (H, W, 3) uint8 RGB images (src2 entries or src2 itself may be NULL: no mixup);
 * hw (B, 2) int32 source sizes, flip (B) int32, trans_inv (B, 6) float64 (output -> source), all on the device.
 * mode 0: out (B, 3, out_h, out_w) fp32 = the reference's tensor; mode 1: out (B, out_h, out_w, 3) uint8 (feeds y3d_stem_im2col_u8,
 * which divides by 255 itself). */
int y3d_kitti_image_aug(const unsigned char* const* src, const unsigned char* const* src2, const int* hw, const int* flip, const double* trans_inv,
                        int B, int out_h, int out_w, int mode, void* out, void* stream);
/* One-to-many depth fusion of the 3D validator — YOLOv10_3DDetectionValidator.aggregate_o2m_preds models/yolov10_3D/val.py:78-102:
 * predsO (B, K, C), predsM (B, KM, C) post-processed rows [xyxy | ... | depth (C-4) | depth log-variance (C-3) | score | label (C-1)];
 * the one-to-many rows with IoU > iou_thres, the same label and exp(-log-variance) > thres vote on the depth through a weighted
 * gaussian kernel density (silverman bandwidth as scikit-learn defines it), evaluated at nprop float32 proposals between the
 * extreme votes; out (B, K, C) = predsO with the most likely proposal as depth. */
int y3d_kde_depth_fusion(const float* predsO, int B, int K, const float* predsM, int KM, int C, float thres, float iou_thres, int nprop,
                         float* out, void* stream);
/* KITTI decode of the post-processed detections (data/datasets/kitti.py:519-576 `decode_preds`, called by the validator's
 * `_prepare_preds`, models/yolov10_3D/val.py:210-214).  preds (B, K, 37) fp32 rows [xyxy | centre-3d | size residual | 24 heading |
 * depth | depth log-variance | score logit | label]; calib (B, 6) = (cu, cv, fu, fv, tx, ty) of the original image; ratio (B, 2) =
 * ratio_pad[i][0]; inv_trans (B, 2, 3) or NULL (undo_augment = False: the fixed 1242/1280, 375/384 rescale); mean_size (nc, 3).
 * out (B, K, 14) f64 [cls, alpha, x1, y1, x2, y2, h, w, l, x, y, z, ry, score]; keep (B, K) u8 = !(score < threshold). */
int y3d_kitti_decode(const float* preds, int B, int K, const double* calib, const double* ratio, const double* inv_trans,
                     const double* mean_size, int nc, int use_camera_dis, double threshold, double* out, unsigned char* keep, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer step as multi-tensor launches (optim.hip): clip_grad_norm_ + SGD(nesterov, weight decay) — engine/trainer.py:567-575,
 * 734-790.  All table arguments are DEVICE arrays: tensor t has sizes[t] fp32 elements at param_ptrs[t] / grad_ptrs[t] / buf_ptrs[t];
 * workgroup c handles elements [chunk_off[c]*chunk, +chunk) of tensor chunk_tensor[c].
 * ---------------------------------------------------------------------------------------------- */
int y3d_mt_sqnorm(const int64_t* grad_ptrs, const int64_t* sizes, const int* chunk_tensor, const int* chunk_off, int nchunks, int chunk,
                  float* partials, void* stream);
/* dst_t = src_t * scale for every tensor of the table (same table layout): the step's gradient tensors -> the flat buffer that is
 * all-reduced over RCCL (ddp.FlatGradReducer; reference: DistributedDataParallel's gradient buckets, engine/trainer.py:225-236) */
int y3d_mt_copy(const int64_t* src_ptrs, const int64_t* dst_ptrs, const int64_t* sizes, const int* chunk_tensor, const int* chunk_off,
                int nchunks, int chunk, float scale, void* stream);
/* out[0] = total gradient L2 norm, out[1] = min(1, max_norm / (norm + 1e-6)) */
int y3d_mt_clip_coef(const float* partials, int nchunks, float max_norm, float* out_norm_clip, void* stream);
/* g = grad*clip (+ wd*p); buf = first ? g : momentum*buf + g; p -= lr * (nesterov ? g + momentum*buf : buf) */
int y3d_mt_sgd(const int64_t* param_ptrs, const int64_t* grad_ptrs, const int64_t* buf_ptrs, const int64_t* sizes, const float* lr,
               const float* wd, const int* chunk_tensor, const int* chunk_off, int nchunks, int chunk, float momentum, int nesterov,
               int first_step, const float* norm_clip, void* stream);

/* torch.optim.AdamW step (amsgrad off) — the reference's optimizer for short schedules (engine/trainer.py:752-764, `optimizer: auto`):
 * g = grad*clip; p *= 1 - lr*wd; m += (1-beta1)(g - m); v = beta2 v + (1-beta2) g^2; p -= lr/bias_corr1 * m / (sqrt(v)/bias_corr2_sqrt + eps) */
int y3d_mt_adamw(const int64_t* param_ptrs, const int64_t* grad_ptrs, const int64_t* m_ptrs, const int64_t* v_ptrs, const int64_t* sizes,
                 const float* lr, const float* wd, const int* chunk_tensor, const int* chunk_off, int nchunks, int chunk, float beta1, float beta2,
                 float eps, float bias_corr1, float bias_corr2_sqrt, const float* norm_clip, void* stream);
/* ModelEMA.update (utils/torch_utils.py:431-443, called from optimizer_step engine/trainer.py:574-575): e = e*decay + (1-decay)*m over
 * every floating-point state_dict tensor, reps[t] times for tensor t (the reference walks state_dict KEYS, and the aliased one-to-one
 * head branches appear under two keys each).  guard: NULL, or y3d_mt_clip_coef's array — the update is then a no-op when
 * guard[2] == 0 (it follows an optimizer step that was skipped) */
int y3d_mt_ema(const int64_t* ema_ptrs, const int64_t* model_ptrs, const int64_t* sizes, const int* reps, const int* chunk_tensor,
               const int* chunk_off, int nchunks, int chunk, float decay, float one_minus_decay, const float* guard, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* Y3D_H */
