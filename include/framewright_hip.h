/*
 * framewright_hip.h — C-ABI of libframewright_hip.so, the MI355X (gfx950 / CDNA4) drop-in for the
 * conv-net hot path of FrameWright (Real-ESRGAN upscale, NAFNet temporal denoise, RIFE interpolation).
 *
 * The reference (/root/reference, pure Python) has no native boundary of its own: the arithmetic is
 * delegated to third-party torch modules / external binaries.  Each entry point below therefore names the
 * reference call it replaces (file:line relative to the reference tree), and INTEGRATION.md shows the
 * ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns an int status: 0 = ok, 1 = invalid argument, 2 = GPU out of memory
 *     (message contains "memory", so restorer.py:1746's tile-downshift retry still triggers),
 *     3 = HIP runtime error, 4 = internal error.  fw_last_error() returns the message for the calling thread.
 *   - plain pointers and sizes only; buffers are caller-owned.  `loc` arguments say where a buffer lives:
 *     FW_HOST (pageable or pinned host memory) or FW_DEVICE (hipMalloc'd / torch.cuda memory on the
 *     handle's device).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are asynchronous on
 *     that stream when every buffer is FW_DEVICE; calls with FW_HOST buffers return after the copy back.
 *   - one handle per GPU; calls on one handle are serialised by an internal mutex, so the reference's
 *     ThreadPoolExecutor callers (restorer.py:1894) may share a handle.  A handle owns ONE workspace: when two
 *     FW_DEVICE calls on it are enqueued on different streams, the second stream waits (hipStreamWaitEvent) for an
 *     event the first call recorded behind its last kernel, so the device work of a handle never overlaps whatever
 *     the streams; callers that WANT two forwards in flight create one handle per stream.
 *   - images are H x W x 3 uint8 in BGR order (cv2 convention, plugins/base.py:186-250).
 */
#ifndef FRAMEWRIGHT_HIP_H
#define FRAMEWRIGHT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FW_OK 0
#define FW_ERR_INVALID 1
#define FW_ERR_OOM 2
#define FW_ERR_HIP 3
#define FW_ERR_INTERNAL 4

#define FW_HOST 0
#define FW_DEVICE 1

/* operand (MFMA input / activation storage) type; accumulation is always fp32 */
#define FW_DTYPE_BF16 0
#define FW_DTYPE_F16 1

typedef struct fw_rrdbnet fw_rrdbnet;
typedef struct fw_nafnet fw_nafnet;

/* -------------------------------------------------------------------------------------------------
 * Library
 * ------------------------------------------------------------------------------------------------- */

/* Message of the last failing call on this thread ("" if none).  Never NULL. */
const char* fw_last_error(void);

/* ABI version of this header: 3.  Bumped whenever entry points are added or changed (1 -> 2 and 2 -> 3 were additive: a binder of
 * version 1 or 2 keeps working against this library). */
int fw_abi_version(void);

/* Number of visible HIP devices (0 when there is no GPU; never fails). */
int fw_device_count(void);

/* -------------------------------------------------------------------------------------------------
 * Real-ESRGAN: RRDBNet generator
 * replaces  basicsr RRDBNet construction + RealESRGANer(...) in get_upsampler()
 *           (processors/pytorch_realesrgan.py:85-173) and upsampler.enhance() (:223,:227;
 *           processors/enhancement/super_resolution.py:524).
 * ------------------------------------------------------------------------------------------------- */

/* Create an RRDBNet(num_in_ch=3, num_out_ch=3, num_feat=64, num_block, num_grow_ch=32, scale) on
 * `device_id`.  scale in {2,4} (pytorch_realesrgan.py:103-129: x4plus nb=23, anime_6B nb=6, x2plus nb=23). */
int fw_rrdbnet_create(int device_id, int num_block, int scale, int dtype, fw_rrdbnet** out);

/* Upload one convolution.  `key` is the BasicSR state-dict prefix ("conv_first", "body.7.rdb2.conv4",
 * "conv_body", "conv_up1", "conv_up2", "conv_hr", "conv_last"); weight is torch layout
 * [cout][cin][3][3] fp32 (host memory), bias [cout] fp32.  Weights are converted to the operand type
 * (round to nearest even) and packed into MFMA fragments. */
int fw_rrdbnet_set_conv(fw_rrdbnet* net, const char* key, const float* weight, const float* bias, int cout,
                        int cin);

/* Verify that every convolution of the architecture has been uploaded. */
int fw_rrdbnet_finalize(fw_rrdbnet* net);

/* One frame: uint8 BGR H x W x 3 -> uint8 BGR (scale*H) x (scale*W) x 3.
 * Equivalent to RealESRGANer.enhance(img, outscale=scale) for a 3-channel uint8 image with tile=0,
 * pre_pad=0: BGR->RGB, /255, (x2 model: reflect mod-pad to even + pixel_unshuffle(2)), RRDBNet forward,
 * clamp(0,1), *255, round-half-even, RGB->BGR.  If out_rgb_f32 is non-NULL (device memory,
 * (scale*H) x (scale*W) x 3 floats, RGB order) the un-clamped network output is stored there as well
 * (used by the parity tests to measure max-abs before quantisation).  out_bgr may be NULL then. */
int fw_rrdbnet_upscale_u8(fw_rrdbnet* net, const uint8_t* in_bgr, int in_loc, int height, int width,
                          uint8_t* out_bgr, int out_loc, float* out_rgb_f32, void* stream);

/* The same for a 16-bit frame (uint16 BGR, range 65535): RealESRGANer.enhance takes this branch when the image maximum
 * exceeds 256 (cv2.imread(IMREAD_UNCHANGED) of a 16-bit PNG, reference pytorch_realesrgan.py:200-227): /65535 on the way in,
 * clamp(0,1) * 65535, round-half-even, uint16 on the way out. */
int fw_rrdbnet_upscale_u16(fw_rrdbnet* net, const uint16_t* in_bgr, int in_loc, int height, int width,
                           uint16_t* out_bgr, int out_loc, float* out_rgb_f32, void* stream);

/* Bytes of device workspace the net needs for an H x W input (0 on invalid arguments). */
size_t fw_rrdbnet_workspace_bytes(const fw_rrdbnet* net, int height, int width);

/* Algorithmic FLOPs (2 x MAC, un-padded channel counts) of one forward on an H x W input. */
double fw_rrdbnet_flops(const fw_rrdbnet* net, int height, int width);

/* Per-launch timing: when enabled, every conv launch of subsequent upscale calls is bracketed by HIP events on
 * the launch stream.  fw_rrdbnet_profile_read synchronises the stream and returns the number of conv
 * launches timed since the last read, their summed duration (ms) and summed algorithmic FLOPs. */
int fw_rrdbnet_profile_enable(fw_rrdbnet* net, int on);
int fw_rrdbnet_profile_read(fw_rrdbnet* net, int* launches, double* total_ms, double* total_flops);

/* Release device memory.  NULL is allowed. */
int fw_rrdbnet_destroy(fw_rrdbnet* net);

/* -------------------------------------------------------------------------------------------------
 * Operator-level entry points (used by the parity tests; the model calls above use the same kernels).
 * All pointers are device memory on `device_id`'s current context; tensors are NHWC.
 * ------------------------------------------------------------------------------------------------- */

/* Pack a torch-layout conv weight [cout][cin][3][3] (host fp32) into MFMA fragments (host uint16 buffer).
 * Returns the number of uint16 elements (call with dst = NULL to size the buffer).  cout is padded to
 * 32*cout_tiles, cin to 32*cin_chunks. */
size_t fw_pack_conv3x3(int dtype, const float* weight, int cout, int cin, int cout_tiles, int cin_chunks,
                       uint16_t* dst);

/* y = act(conv3x3(x) + bias) written to channels [out_coff, out_coff + 32*cout_tiles) of an NHWC operand-typed
 * buffer.  x: operand-typed; pixel (y,x) of 32-channel chunk c starts at element
 * c*in_plane_stride + (y*W + x)*in_cstride.  in_plane_stride = 0 (or 32) is plain interleaved NHWC with in_cstride
 * channels per pixel; in_plane_stride = H*W*in_cstride with in_cstride = 32 is the chunk-planar layout the RRDB
 * trunk uses (every chunk read is a contiguous stream of whole cache lines).  The first 32*cin_chunks channels are
 * contracted.  The two 32-channel halves of a 64-channel output are out_plane_stride elements apart (0 = 32).
 * upsample2x = 1: x is (H/2 x W/2) and is nearest-neighbour upsampled on the fly (conv_up1/conv_up2 of
 * aesrgan_face.py:258-266).  res1/res2 (fp32 NHWC, 32*cout_tiles channels, may be NULL):
 *   res1 == NULL:  y = act(acc + bias)
 *   res1 != NULL:  y = (acc + bias) * s1 + res1; if res2: y = y * s2 + res2     (aesrgan_face.py:189,204)
 * out / out_f32 may each be NULL. */
int fw_conv3x3_nhwc(int dtype, const void* x, int in_cstride, long in_plane_stride, int cin_chunks, int height,
                    int width, const void* packed_weight, const float* bias, int cout_tiles, int act_lrelu,
                    int upsample2x, const float* res1, float s1, const float* res2, float s2, void* out,
                    int out_cstride, long out_plane_stride, int out_coff, float* out_f32, void* stream);

/* Same, with the extras the IFNet blocks need: chan_scale (fp32 [32*cout_tiles] or NULL) multiplies (acc + bias) per
 * output channel before the residual add, post_act = 1 applies LeakyReLU(0.2) AFTER the residual add
 * (ResConv: lrelu(conv(x) * beta + x)), and f32_cstride / f32_coff place res1 / res2 / out_f32 in a
 * channel slice of a wider fp32 NHWC buffer (0 / 0 = a dense 32*cout_tiles-channel buffer). */
int fw_conv3x3_nhwc_ex(int dtype, const void* x, int in_cstride, long in_plane_stride, int cin_chunks, int height,
                       int width, const void* packed_weight, const float* bias, int cout_tiles, int act_lrelu,
                       int upsample2x, const float* res1, float s1, const float* res2, float s2, const float* chan_scale,
                       int post_act, int f32_cstride, int f32_coff, void* out, int out_cstride, long out_plane_stride,
                       int out_coff, float* out_f32, void* stream);

/* act_lrelu == 2 in fw_conv3x3_nhwc_ex (res1 == NULL): PReLU, y = max(t, 0) + chan_scale[n] * min(t, 0) with t = acc + bias —
 * the activation of SRVGGNetCompact (realesr-animevideov3, realesr-general-x4v3). */

/* uint8 BGR H x W x 3 -> operand-typed [H][W][out_cstride] RGB/255 in channels 0..2, zeros above (out_cstride >= 32).
 * replaces  img.astype(float32)/255 + BGR->RGB + HWC->CHW of RealESRGANer.pre_process (third-party; call site
 *           processors/pytorch_realesrgan.py:223). */
int fw_u8_to_nhwc(int dtype, const uint8_t* in_bgr, int height, int width, void* out, int out_cstride, void* stream);

/* SRVGGNetCompact tail: out = PixelShuffle(scale)(conv) + nearest-upsample(input), conv = fp32 [H][W][conv_cstride] with
 * channel c*scale^2 + i*scale + j; RGB float [scale*H][scale*W][3] and/or clamp -> x255 -> rint -> uint8 BGR.
 * replaces  the tail of the SRVGGNetCompact forward (third-party `realesrgan.archs.srvgg_arch`; the reference declares
 *           these checkpoints at processors/pytorch_realesrgan.py:119-128). */
int fw_pixel_shuffle_add_u8(const float* conv, int conv_cstride, const uint8_t* in_bgr, int height, int width, int scale,
                            uint8_t* out_bgr, float* out_rgb_f32, void* stream);

/* Two chained growth convolutions of a residual dense block in one kernel (aesrgan_face.py:184-187):
 *   x_a = lrelu(conv_a(x[0 : 32*in_chunks]) + bias_a)            (32 channels) -> out_a
 *   x_b = lrelu(conv_b(cat(x[0 : 32*in_chunks], x_a)) + bias_b)  (32 channels) -> out_b
 * packed_weight_a = fw_pack_conv3x3(cout 32, cin 32*in_chunks, 1, in_chunks), packed_weight_b likewise with
 * in_chunks + 1 (its last chunk multiplies x_a).  out_a / out_b: operand-typed, out_cstride elements per pixel.
 * The shared input chunks are fetched from HBM once for both convolutions and x_a feeds conv_b from LDS. */
int fw_conv3x3_pair_nhwc(int dtype, const void* x, int in_cstride, long in_plane_stride, int in_chunks, int height,
                         int width, const void* packed_weight_a, const float* bias_a, const void* packed_weight_b,
                         const float* bias_b, void* out_a, void* out_b, int out_cstride, void* stream);

/* lrelu(conv3x3(nearest_x2(x)) + bias), 64 -> 64 channels (conv_up1 / conv_up2, aesrgan_face.py:258-266) evaluated on the SOURCE
 * grid as four 2x2 phase convolutions with summed weights: output pixel (2y + a, 2x + b) folds its three tap rows / columns onto two
 * source rows / columns, 4 instead of 9 MFMA taps.  fw_pack_conv_up2x_phase packs a torch-layout weight [64][64][3][3] (host fp32;
 * sums in fp64, rounded to the operand type once; returns the number of uint16, dst = NULL to size the buffer).
 * x: operand-typed source image height x width, 64 channels as two 32-channel chunks in_plane_stride elements apart (0 = 32:
 * interleaved NHWC with in_cstride channels per pixel); out: operand-typed [2*height][2*width][out_cstride], its two 32-channel
 * halves out_plane_stride elements apart (0 = 32). */
size_t fw_pack_conv_up2x_phase(int dtype, const float* weight, uint16_t* dst);
int fw_conv_up2x_phase_nhwc(int dtype, const void* x, int in_cstride, long in_plane_stride, int height, int width,
                            const void* packed_weight, const float* bias, int act_lrelu, void* out, int out_cstride,
                            long out_plane_stride, void* stream);

/* -------------------------------------------------------------------------------------------------
 * TAP temporal denoise: NAFNet
 * replaces  basicsr NAFNet construction + load in TAPDenoiser._load_nafnet (processors/tap_denoise.py:335-364),
 *           _preprocess_frame / self._model(tensor) / _postprocess_frame (:373-415, :431-434, :455-459).
 * ------------------------------------------------------------------------------------------------- */

/* NAFNet(img_channel=3, width, middle_blk_num, enc_blk_nums[num_levels], dec_blk_nums[num_levels]); the reference
 * passes width=64, middle=12, enc=[2,2,4,8], dec=[2,2,2,2] (tap_denoise.py:340-346).  width in {32, 64}. */
int fw_nafnet_create(int device_id, int width, int middle_blk_num, const int* enc_blk_nums, const int* dec_blk_nums,
                     int num_levels, int dtype, fw_nafnet** out);

/* Upload one state-dict tensor by its key (host fp32, torch layout, `numel` elements): intro.weight/bias,
 * ending.weight/bias, downs.{l}.weight/bias, ups.{i}.0.weight, {encoders.{l}.{j} | middle_blks.{j} |
 * decoders.{i}.{j}}.{norm1,norm2}.{weight,bias} / conv{1..5}.{weight,bias} / sca.1.{weight,bias} / beta / gamma. */
int fw_nafnet_set_tensor(fw_nafnet* net, const char* key, const float* data, size_t numel);
int fw_nafnet_finalize(fw_nafnet* net);

/* One frame (or tile): uint8 BGR H x W x 3 -> uint8 BGR H x W x 3.  BGR->RGB, /255, zero pad to a multiple of
 * 2^num_levels, NAFNet forward (+ input residual), crop, np.clip(x*255, 0, 255).astype(uint8) — truncation, as
 * tap_denoise.py:412 — RGB->BGR.  out_rgb_f32 (optional, device, H x W x 3 RGB) receives the un-quantised output. */
int fw_nafnet_denoise_u8(fw_nafnet* net, const uint8_t* in_bgr, int in_loc, int height, int width, uint8_t* out_bgr,
                         int out_loc, float* out_rgb_f32, void* stream);
double fw_nafnet_flops(const fw_nafnet* net, int height, int width);
int fw_nafnet_destroy(fw_nafnet* net);

/* -------------------------------------------------------------------------------------------------
 * TAP driver arithmetic on uint8 frames (all pointers device memory; bit-exact restatements)
 * ------------------------------------------------------------------------------------------------- */

/* dst[th][tw][3] = src[y0:y0+th, x0:x0+tw]               (frame[y1:y2, x1:x2], tap_denoise.py:452) */
int fw_u8_crop(const uint8_t* src, int height, int width, int y0, int x0, int th, int tw, uint8_t* dst, void* stream);

/* output[y0:.., x0:..] += tile * tile_weight ; weight[...] += tile_weight, with the linear ramps of
 * tap_denoise.py:461-486 on the tile edges that are interior to the frame (np.linspace(0, 1, overlap)). */
int fw_tile_blend_accumulate(float* acc, float* wsum, int height, int width, const uint8_t* tile, int y0, int x0, int th,
                             int tw, int overlap, void* stream);

/* out = (output / max(weight, 1e-8)).astype(uint8)        (tap_denoise.py:484-486) */
int fw_tile_blend_finish(const float* acc, const float* wsum, int height, int width, uint8_t* out, void* stream);

/* result = sum_k float32(frame_k) * float32(w_k) accumulated in float32, .astype(uint8)   (tap_denoise.py:526-534).
 * `frames` is a HOST array of `count` (<= 16) DEVICE pointers; weights are the normalised host weights. */
int fw_temporal_average_u8(const uint8_t* const* frames, const float* weights, int count, size_t nbytes, uint8_t* out,
                           void* stream);

/* out = (original * (1 - s) + denoised * s).astype(uint8)  (tap_denoise.py:614-618); s is a double like the
 * reference's Python float: (1 - s) is formed in double, then both factors are rounded once to float32. */
int fw_strength_blend_u8(const uint8_t* original, const uint8_t* denoised, double strength, size_t nbytes, uint8_t* out,
                         void* stream);

/* preserve_grain (tap_denoise.py:621-632; motion-adaptive variant :1015-1023 with its own factor):
 *   grain = cv2.subtract(gray(original), cv2.GaussianBlur(gray(original), (0, 0), 3))
 *   out   = cv2.add(denoised, (GRAY2BGR(grain) * factor).astype(np.uint8))
 * on uint8 BGR H x W x 3 device buffers; `scratch` = height*width uint16 of device memory.  OpenCV's 8-bit fixed-point
 * arithmetic (BGR2GRAY 14-bit weights; 19-tap bit-exact Gaussian, BORDER_REFLECT_101; saturating subtract / add). */
int fw_grain_addback_u8(const uint8_t* original, const uint8_t* denoised, int height, int width, double factor,
                        uint16_t* scratch, uint8_t* out, void* stream);

/* cv2.resize(output, (int(w*outscale), int(h*outscale)), interpolation=cv2.INTER_LANCZOS4) on an 8-bit H x W x C image
 * (C <= 4; device pointers): the last step of realesrgan.RealESRGANer.enhance when outscale != netscale — reference call
 * site src/framewright/processors/pytorch_realesrgan.py:223 (`upsampler.enhance(img, outscale=config.scale_factor)`,
 * scale_factor 2 with a x4 model).  OpenCV's fixed-point arithmetic (8x8 taps, 11-bit coefficients, clamped borders);
 * synchronises `stream` before it returns. */
int fw_resize_lanczos4_u8(const uint8_t* src, int src_h, int src_w, int channels, uint8_t* dst, int dst_h, int dst_w,
                          void* stream);
/* The same on 16-bit images: RealESRGANer.enhance with outscale != netscale on a 16-bit frame (cv2.resize of a ushort image runs
 * OpenCV's float path: float weights, eight products summed left to right per pass, saturate_cast<ushort>(cvRound)). */
int fw_resize_lanczos4_u16(const uint16_t* src, int src_h, int src_w, int channels, uint16_t* dst, int dst_h, int dst_w,
                           void* stream);

/* -------------------------------------------------------------------------------------------------
 * The host arithmetic of AESRGANFaceRestorer around its network
 * replaces  `cv2.resize(enhanced_face, (target_w, target_h))` and the feathered float32 blend of `_paste_face_back`,
 *           processors/aesrgan_face.py:543-584 (the network itself: fw_aesrgan_*).  Device pointers.
 * fw_resize_linear_u8: cv2.resize with its default INTER_LINEAR on 8-bit images - OpenCV's fixed-point bilinear (11-bit coefficients,
 * clamped borders), its 2 x 2 "area fast" average for an exact 2:1 decimation (the default upscale_factor), a copy at equal size;
 * synchronises `stream` when it had to build tables.
 * fw_face_paste_u8: frame[y1:y2, x1:x2] = orig * (1 - mask * s) + enhanced * mask * s in float32 with the reference's feather mask
 * (min(w, h) // 8 rows / columns scaled by i / feather), truncating cast; `enhanced` is (y2 - y1) x (x2 - x1) x 3; in place. */
int fw_resize_linear_u8(const uint8_t* src, int src_h, int src_w, int channels, uint8_t* dst, int dst_h, int dst_w, void* stream);
int fw_face_paste_u8(uint8_t* frame, int height, int width, int x1, int y1, int x2, int y2, const uint8_t* enhanced, double strength,
                     void* stream);

/* -------------------------------------------------------------------------------------------------
 * Restormer building blocks (the reference's DEFAULT TAP model)
 * replaces  `Restormer(dim=48, num_blocks=[4,6,6,8], num_refinement_blocks=4, heads=[1,2,4,8],
 *           ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias')` + `self._model(tensor)`
 *           (processors/tap_denoise.py:299-333, 458); architecture per SURVEY.md §A.4; host sequencing
 *           framewright_amd/restormer.py.  Device pointers; the residual stream is fp32 NHWC with a padded channel stride.
 * ------------------------------------------------------------------------------------------------- */

/* LayerNorm over the first `channels` of each pixel, (x - mean) / sqrt(var + eps) * weight + bias (bias may be NULL),
 * fp32 [pixels][x_stride] -> operand-typed [pixels][out_stride]; output channels [channels, zero_to) are zeroed. */
int fw_layernorm_nhwc(int dtype, const float* x, long x_stride, long pixels, int channels, const float* weight,
                      const float* bias, float eps, void* out, long out_stride, int zero_to, void* stream);

/* 1x1 convolution as an MFMA GEMM.  fw_pack_pointwise packs weight [cout][k] (fp32, host; k % 32 == 0) into fragments
 * (returns the uint16 count, dst may be NULL to query).  a: operand-typed or fp32 [pixels][a_stride]; outputs: typed
 * and/or fp32, 32*cout_tiles channels; with res_f32: out_f32 = res_f32 + (acc + bias) * chan_scale (same stride). */
size_t fw_pack_pointwise(int dtype, const float* weight, int cout, int k, void* dst);
int fw_pointwise_nhwc(int dtype, const void* a, int a_is_f32, long a_stride, long pixels, int k, const void* packed_weight,
                      const float* bias, int cout_tiles, void* out_typed, long out_stride, float* out_f32, long f32_stride,
                      const float* res_f32, const float* chan_scale, void* stream);

/* Depthwise 3x3, zero padding, no bias, weight fp32 [channels][9].  mode 0: plain; mode 1: the GDFN gate,
 * out[c] = gelu(dw(x)[c]) * dw(x)[channels/2 + c] for c < channels/2 (exact erf GELU). */
int fw_dwconv3x3_nhwc(int dtype, const void* x, long x_stride, int height, int width, int channels, const float* weight,
                      int mode, void* out, long out_stride, void* stream);

/* MDTA "transposed" attention.  qkv: operand-typed [pixels][stride], q at channel 0, k at k_off, v at v_off, heads*ch
 * channels each (ch = 48 or 96).  fw_attn_matrix: attn[head][c1][c2] = softmax_c2(normalize(q)_c1 . normalize(k)_c2 *
 * temperature[head]) with the dot products over all pixels (deterministic two-level reduction; workspace of
 * fw_attn_workspace_floats floats).  fw_attn_apply: out[p][head*ch + c1] = sum_c2 attn[head][c1][c2] * v[p][head*ch + c2]. */
size_t fw_attn_workspace_floats(int heads, int ch);
int fw_attn_matrix(int dtype, const void* qkv, long stride, long pixels, int k_off, int heads, int ch,
                   const float* temperature, float* workspace, float* attn, void* stream);
/* fw_attn_matrix on the matrix cores (what the engine runs): q and k are first rewritten pixel-major into `qk_scratch`
 * (fw_attn_qk_scratch_elems(pixels, heads, ch) operand-typed elements of device memory), then G = q^T k is an MFMA
 * contraction over the pixel axis; same partial-sum workspace, same deterministic reduction, same result up to fp32
 * summation order. */
size_t fw_attn_qk_scratch_elems(long pixels, int heads, int ch);
int fw_attn_matrix_mfma(int dtype, const void* qkv, long stride, long pixels, int k_off, int heads, int ch,
                        const float* temperature, float* workspace, void* qk_scratch, float* attn, void* stream);
int fw_attn_apply(int dtype, const void* qkv, long stride, long pixels, int v_off, int heads, int ch, const float* attn,
                  void* out, long out_stride, int zero_to, void* stream);
/* The attention matrices as one block-diagonal [k_pad x k_pad] weight in fw_pack_pointwise's fragment order (device buffer of
 * fw_pack_pointwise(dtype, NULL, k_pad, k_pad, NULL) uint16), so that attn @ v runs through fw_pointwise_nhwc on the matrix
 * cores — what the engine uses; fw_attn_apply is the plain form of the same product. */
int fw_attn_pack(int dtype, const float* attn, int heads, int ch, int k_pad, void* packed, void* stream);
/* project_out folded into the attention (restormer.py / reference Attention.forward: project_out(attn @ v)): packed =
 * (proj_weight [dim][dim] fp32) x blockdiag(attn) in fw_pack_pointwise's layout with cout_tiles 32-row tiles, so that
 * project_out(attn @ v) + x is one fw_pointwise_nhwc over v with the residual epilogue. */
int fw_attn_proj_pack(int dtype, const float* attn, const float* proj_weight, int heads, int ch, int k_pad, int cout_tiles, void* packed,
                      void* stream);

/* AESRGAN's AttentionBlock (reference src/framewright/processors/aesrgan_face.py:142-168, the in-tree net behind
 * AESRGANFaceRestorer): attention = softmax(q^T k) over ALL pixels, out = gamma * (v @ attention^T) + x.
 * fw_attn_softmax_rows: p[i][j] = softmax_j(sum_{c<d} q[i][c] k[j][c]) as an operand-typed [pixels][p_stride] matrix, columns
 * [pixels, p_stride) zeroed (p_stride = pixels padded to 32).  fw_pack_pointwise_transposed: fw_pack_pointwise's fragments
 * for the weight W[co][kk] = src[kk][co] from a typed device matrix (W = v^T), so that out = x + gamma * (p @ v) is
 * fw_pointwise_nhwc(a = p, k = p_stride, res_f32 = x, chan_scale = gamma). */
int fw_attn_softmax_rows(int dtype, const void* q, long q_stride, const void* k, long k_stride, long pixels, int d, void* p,
                         long p_stride, void* stream);
int fw_pack_pointwise_transposed(int dtype, const void* src, long src_stride, long k_valid, int cout, int k_pad, void* packed,
                                 void* stream);

/* torch.nn.PixelShuffle(2) (unshuffle = 0) / PixelUnshuffle(2) (unshuffle = 1) on fp32 NHWC; low_h x low_w is the
 * low-resolution size, `channels` the channel count at HIGH resolution; dst channels start at dst_coff. */
int fw_pixel_shuffle2_f32(const float* src, long src_stride, int low_h, int low_w, int channels, float* dst,
                          long dst_stride, int dst_coff, int unshuffle, void* stream);
/* dst[p][dst_coff + c] = src[p][c], c < channels (torch.cat along channels). */
int fw_copy_channels_f32(const float* src, long src_stride, long pixels, int channels, float* dst, long dst_stride,
                         int dst_coff, void* stream);
/* fp32 [pixels][channels] (channels % 32 == 0) -> operand-typed chunk-planar [channels/32][pixels][32]. */
int fw_f32_to_planar(int dtype, const float* x, long pixels, int channels, void* out, void* stream);
/* out = clip((rgb + input/255) * 255, 0, 255).astype(uint8), RGB -> BGR (tap_denoise.py:399-415: truncation); rgb is fp32
 * [H][padded_w][rgb_cstride] with R,G,B in channels 0..2. */
int fw_tap_post_u8(const uint8_t* in_bgr, const float* rgb, int height, int width, int padded_width, int rgb_cstride,
                   uint8_t* out_bgr, float* out_rgb_f32, void* stream);

/* -------------------------------------------------------------------------------------------------
 * Classical motion-compensated temporal denoise (device pointers)
 * replaces  OpticalFlowEstimator.warp_frame (processors/temporal_denoise.py:440-477) and the accumulation of
 *           TemporalDenoiser._denoise_with_flow / _denoise_simple (:1521-1605).  The dense flow (cv2 Farneback / DIS) is
 *           estimated on the host and is not part of this path.
 * ------------------------------------------------------------------------------------------------- */

/* accumulated[p] += aligned[p] * w[p]; weight_sum[p] += w[p]  (float64, like the reference), with
 *   aligned = flow ? cv2.remap(frame, x +/- flow_x, y +/- flow_y, INTER_LINEAR, BORDER_REFLECT_101) : frame
 *   w       = weight_scale * (weight_map ? weight_map[p] : 1), halved where magnitude[p] > motion_threshold.
 * flow_x / flow_y / weight_map / magnitude: fp32 [H][W] or NULL; inverse != 0 subtracts the flow. */
int fw_flow_accumulate_u8(const uint8_t* frame_bgr, const float* flow_x, const float* flow_y, const float* weight_map,
                          double weight_scale, const float* magnitude, float motion_threshold, int inverse, int height,
                          int width, double* accumulated, double* weight_sum, void* stream);
/* out = (accumulated / max(weight_sum, 1e-6)).astype(uint8) */
int fw_flow_accumulate_finish_u8(const double* accumulated, const double* weight_sum, int height, int width,
                                 uint8_t* out_bgr, void* stream);

/* -------------------------------------------------------------------------------------------------
 * RIFE frame interpolation: IFNet v4.6 building blocks (device pointers, fp32 NHWC small-channel tensors)
 * replaces  the arithmetic inside the external binary the reference shells out to,
 *           `rife-ncnn-vulkan -m rife-v4.6` (processors/interpolation.py:628-650); architecture per SURVEY.md §A.5.
 *           The convolutions of the IFBlocks run through fw_conv3x3_nhwc_ex; the host sequencing is
 *           framewright_amd/rife.py (IFNetEngine).
 * ------------------------------------------------------------------------------------------------- */

/* uint8 BGR H x W x 3 -> fp32 RGB/255 [padded_h][padded_w][3], zero outside H x W. */
int fw_u8_to_rgb_f32(const uint8_t* in_bgr, int height, int width, int padded_height, int padded_width, float* out,
                     void* stream);

/* dst[y][x][dst_coff + c] = mul * F.interpolate(src, scale_factor, "bilinear", align_corners=False)[c][y][x]. */
int fw_resize_bilinear_f32(const float* src, int src_h, int src_w, int channels, float* dst, int dst_h, int dst_w,
                           int dst_cstride, int dst_coff, float scale_factor, float mul, void* stream);

/* x = cat(warp(img0, flow[:2]), warp(img1, flow[2:4]), timestep, mask) (8 ch) — or cat(img0, img1, timestep) (7 ch) when
 * flow == mask == NULL.  warp = grid_sample(bilinear, border, align_corners=True) with the flow in pixels. */
int fw_ifnet_build_x(const float* img0, const float* img1, const float* flow, const float* mask, int height, int width,
                     float timestep, float* x, void* stream);

/* pixel_unshuffle(2) + cast: src [h][w][src_cstride] (fp32 if src_is_f32, else operand-typed; first `channels` used) ->
 * dst operand-typed [h/2][w/2][dst_channels], channel c*4 + dy*2 + dx, zero above 4*channels.  Front end of the
 * stride-2 convolutions, which run as 3x3 convolutions on the unshuffled tensor. */
int fw_unshuffle2_cast(int dtype, const void* src, int src_is_f32, int height, int width, int channels, int src_cstride,
                       void* dst, int dst_channels, void* stream);

/* [h][w][>=96] with channel ((c6*4 + qy*2 + qx)*4 + py*2 + px) -> [4h][4w][6]: ConvTranspose2d(4,2,1) evaluated as a
 * 3x3 conv with 4 output parities, followed by PixelShuffle(2). */
int fw_depth_to_space4_f32(const float* src, int height, int width, int src_cstride, float* dst, void* stream);

/* flow (+)= bilinear_up(tmp)[:4] * scale ; mask (+)= bilinear_up(tmp)[4]   (first != 0: assign). */
int fw_ifnet_accumulate(const float* tmp, int tmp_h, int tmp_w, int height, int width, float scale, float* flow, float* mask,
                        int first, void* stream);

/* merged = warp(img0, flow[:2]) * sigmoid(mask) + warp(img1, flow[2:4]) * (1 - sigmoid(mask)), cropped to H x W;
 * out_bgr = round_half_even(clamp(merged, 0, 1) * 255) (BGR), out_rgb_f32 = merged (RGB); either may be NULL. */
int fw_ifnet_blend(const float* img0, const float* img1, const float* flow, const float* mask, int padded_height,
                   int padded_width, int height, int width, uint8_t* out_bgr, float* out_rgb_f32, void* stream);

/* Motion-blur reduction of the interpolator (reference interpolation.py:403-455: PIL ImageFilter.UnsharpMask(radius, percent,
 * threshold)) on a uint8 H x W x C device image, bit-exact with Pillow's libImaging (BoxBlur.c / UnsharpMask.c):
 *   blur = `passes` extended-box passes along x, then along y, every pass rounded to 8 bits:
 *          (ww * sum_{|k| <= box_radius} in[i+k] + fw_weight * (in[i-box_radius-1] + in[i+box_radius+1]) + 2^23) >> 24,
 *          edge-replicated (the host derives box_radius / ww / fw_weight from the Gaussian radius exactly as Pillow does);
 *   out  = |in - blur| > threshold ? clip8(in + (in - blur) * percent / 100) : in       (C integer division).
 * scratch_a / scratch_b: two H*W*C-byte device buffers.  `out` may alias `src`. */
int fw_unsharp_mask_u8(const uint8_t* src, int height, int width, int channels, int box_radius, unsigned ww, unsigned fw_weight,
                       int passes, int percent, int threshold, uint8_t* scratch_a, uint8_t* scratch_b, uint8_t* out, void* stream);

/* ---- AESRGAN (RRDB trunk + self-attention blocks) as one engine (csrc/aesrgan.hip) -------------------------------------------------
 * The reference's in-tree network of AESRGANFaceRestorer (processors/aesrgan_face.py:205-269, AttentionBlock :142-168): create,
 * hand over the tensors (BasicSR's names for the trunk / tail: conv_first, body.{i}.rdb{1,2,3}.conv{1..5}, conv_body, conv_up1
 * [, conv_up2], conv_hr, conv_last with .weight / .bias; attn.{i}.query|key|value.weight / .bias and attn.{i}.gamma for the block
 * behind RRDB i), finalize, then fw_aesrgan_forward_rgb on fp32 RGB crops in [0, 1] on the device: what AESRGAN.forward returns for
 * a 1 x 3 x H x W input (NHWC, un-clamped).  The attention matrix is pixels x pixels: crops of up to 512 x 512. */
typedef struct fw_aesrgan fw_aesrgan;
int fw_aesrgan_create(int device_id, int num_block, int scale, int num_attention, int dtype, fw_aesrgan** out);
int fw_aesrgan_set_tensor(fw_aesrgan* net, const char* key, const float* data, size_t numel);
int fw_aesrgan_finalize(fw_aesrgan* net);
int fw_aesrgan_forward_rgb(fw_aesrgan* net, const float* x_rgb, int height, int width, float* out_rgb, void* stream);
size_t fw_aesrgan_workspace_bytes(fw_aesrgan* net, int height, int width);
int fw_aesrgan_destroy(fw_aesrgan* net);

/* ---- SRVGGNetCompact as one engine (csrc/srvgg.hip) ---------------------------------------------------------------------------
 * The network of the Real-ESRGAN checkpoints realesr-animevideov3 (num_conv 16) / realesr-general-x4v3 (num_conv 32), which the
 * reference lists in its model table (processors/pytorch_realesrgan.py:119-128): create, hand over the tensors of the published
 * state dict (body.{2i}.weight [cout][cin][3][3], body.{2i}.bias, body.{2i+1}.weight = PReLU slopes [64]), finalize, then any number
 * of fw_srvgg_upscale_u8 calls (uint8 BGR H x W x 3 in, uint8 BGR sH x sW x 3 and / or RGB float out; FW_HOST or FW_DEVICE buffers).
 * One handle per GPU; calls on a handle are serialised by an internal mutex. */
typedef struct fw_srvgg fw_srvgg;
int fw_srvgg_create(int device_id, int num_feat, int num_conv, int upscale, int dtype, fw_srvgg** out);
int fw_srvgg_set_tensor(fw_srvgg* net, const char* key, const float* data, size_t numel);
int fw_srvgg_finalize(fw_srvgg* net);
int fw_srvgg_upscale_u8(fw_srvgg* net, const uint8_t* in_bgr, int in_loc, int height, int width, uint8_t* out_bgr, int out_loc,
                        float* out_rgb_f32, void* stream);
/* 16-bit frames (RealESRGANer.enhance: max_range 65535): uint16 BGR in, / 65535; uint16 BGR out, x 65535, round. */
int fw_srvgg_upscale_u16(fw_srvgg* net, const uint16_t* in_bgr, int in_loc, int height, int width, uint16_t* out_bgr, int out_loc,
                         float* out_rgb_f32, void* stream);
size_t fw_srvgg_workspace_bytes(const fw_srvgg* net, int height, int width);
double fw_srvgg_flops(const fw_srvgg* net, int height, int width);
int fw_srvgg_destroy(fw_srvgg* net);

/* ---- IFNet v4.6 (RIFE x2 interpolation) as one engine ------------------------------------------------------------------------
 * Replaces the reference's `rife-ncnn-vulkan` subprocess (reference src/framewright/processors/interpolation.py:628-650; model
 * directory `rife-v4.6`, :106-124).  Same life cycle as fw_rrdbnet / fw_nafnet: create, set every tensor of the Practical-RIFE
 * IFNet_HDv3 state dict (fp32, PyTorch layouts; keys `block{0..3}.conv0.{0,1}.0.{weight,bias}`,
 * `block{i}.convblock.{0..7}.conv.{weight,bias}`, `block{i}.convblock.{j}.beta`, `block{i}.lastconv.0.{weight,bias}`),
 * finalize (weight transforms + MFMA fragment packing on the host, upload), then any number of fw_ifnet_interp_u8 calls.
 * One handle per GPU; calls on a handle are serialised by an internal mutex.  FW_IFNET_GRAPH=1 in the environment replays a
 * captured hipGraph when a call repeats (frame size, timestep, buffer addresses). */
typedef struct fw_ifnet fw_ifnet;
int fw_ifnet_create(int device_id, int dtype, fw_ifnet** out);
int fw_ifnet_set_tensor(fw_ifnet* net, const char* key, const float* data, size_t numel);
int fw_ifnet_finalize(fw_ifnet* net);
/* frame0 / frame1: uint8 BGR H x W x 3 (both FW_HOST or both FW_DEVICE); timestep in (0, 1) (0.5 = the mid frame of a x2 pass);
 * out_bgr = round_half_even(clamp(merged, 0, 1) * 255) BGR and/or out_rgb_f32 = merged RGB float (device), either may be NULL.
 * Asynchronous on `stream` for device buffers; host outputs are complete on return. */
int fw_ifnet_interp_u8(fw_ifnet* net, const uint8_t* frame0, const uint8_t* frame1, int in_loc, int height, int width,
                       float timestep, uint8_t* out_bgr, int out_loc, float* out_rgb_f32, void* stream);
size_t fw_ifnet_workspace_bytes(const fw_ifnet* net, int height, int width);
double fw_ifnet_flops(const fw_ifnet* net, int height, int width);
int fw_ifnet_destroy(fw_ifnet* net);

/* ---- Restormer (the reference's default TAP model) as one engine ---------------------------------------------------------------
 * Replaces `basicsr.archs.restormer_arch.Restormer(inp_channels=3, out_channels=3, dim=48, num_blocks=[4,6,6,8],
 * num_refinement_blocks=4, heads=[1,2,4,8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias')` as constructed
 * and called at reference src/framewright/processors/tap_denoise.py:299-333 and :458, together with the pre / post-processing of
 * :373-415 (BGR uint8 in, `np.clip(x * 255, 0, 255).astype(uint8)` out - truncation).  Same life cycle as fw_nafnet: create,
 * set every tensor of the state dict (fp32, PyTorch layouts and key names), finalize, denoise.  One handle per GPU, calls
 * serialised by an internal mutex; the workspace is one arena sized by a dry run of the launch sequence. */
typedef struct fw_restormer fw_restormer;
int fw_restormer_create(int device_id, int dim, const int* num_blocks /* [4] */, int num_refinement_blocks,
                        const int* heads /* [4] */, double ffn_expansion_factor, int dtype, fw_restormer** out);
int fw_restormer_set_tensor(fw_restormer* net, const char* key, const float* data, size_t numel);
int fw_restormer_finalize(fw_restormer* net);
/* in_bgr / out_bgr: uint8 BGR H x W x 3, H and W multiples of 8; out_rgb_f32 (device, optional): the un-quantised RGB output. */
int fw_restormer_denoise_u8(fw_restormer* net, const uint8_t* in_bgr, int in_loc, int height, int width, uint8_t* out_bgr,
                            int out_loc, float* out_rgb_f32, void* stream);
size_t fw_restormer_workspace_bytes(fw_restormer* net, int height, int width);   /* 0 before finalize */
int fw_restormer_destroy(fw_restormer* net);

/* `TemporalDenoiser._preserve_edges` (reference src/framewright/processors/temporal_denoise.py:1636-1667): the Canny edges of
 * `original` (cv2.Canny(gray, t, 3t) on cv2.cvtColor(BGR2GRAY)), dilated 3x3 and blurred (GaussianBlur((5, 5), 0) of edges / 255)
 * into a float mask, blend `original` over `denoised`: out = (original * mask + denoised * (1 - mask)).astype(uint8).  uint8 BGR
 * H x W x 3 device buffers; `scratch`: fw_preserve_edges_scratch_bytes(height, width) bytes of device memory.  OpenCV's 8-bit
 * algorithms, restated (oracle/temporal_ref.py).  The hysteresis iterates to its fixed point: the call synchronises `stream`. */
size_t fw_preserve_edges_scratch_bytes(int height, int width);
int fw_preserve_edges_u8(const uint8_t* original, const uint8_t* denoised, int height, int width, double low_threshold,
                         double high_threshold, void* scratch, uint8_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FRAMEWRIGHT_HIP_H */
