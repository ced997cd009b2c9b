// Internal declarations shared by the HIP translation units of libframewright_hip.so.
// gfx950 (MI355X / CDNA4) only.  Nothing in here is part of the C-ABI; see include/framewright_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <stdexcept>

namespace fw {

// Operand (MFMA A/B and activation storage) type.  Accumulation is always fp32.
enum DType : int { DT_BF16 = 0, DT_F16 = 1 };

// ---- conv3x3 (stride 1, zero pad 1) implicit GEMM on MFMA ---------------------------------
// Input : NHWC, operand-typed, `in_cstride` channels per pixel, the conv reads the first
//         32*cin_chunks channels (a prefix of the residual-dense "concat" buffer).
// Output: depends on the epilogue.
enum ConvEpilogue : int {
    EPI_STORE = 0,     // y = act(acc + bias)                    -> typed NHWC slice (+ optional fp32 copy)
    EPI_RESIDUAL = 1,  // y = (acc+bias)*s1 + res1 [; y = y*s2 + res2] -> fp32 trunk + typed NHWC slice
    EPI_IMAGE = 2,     // 3 output channels: RGB float and/or clamp->x255->rint->uint8 BGR
    // The RRDB trunk kept as a pair of operand-typed tensors instead of fp32: hi = T(x) (the typed planes every conv reads
    // anyway) and lo = T(x - hi), both chunk-planar.  hi + lo carries 16 (bf16) / 22 (f16) mantissa bits.  The residual
    // add runs on the matrix cores: while it contracts conv chunk c the kernel also loads residual plane c (c < n_id)
    // straight into B-fragment registers and accumulates it as id_scale[c] * I (exact products, fp32 accumulate), so the
    // epilogue loads nothing:
    //   y = s1 * (acc + bias + in_id_scale * in[0:64] + sum_c id_scale[c] * plane_c)  ->  out (hi planes) + out_lo (lo planes)
    // 64 output channels; residual plane c feeds output channels [32*(c&1), 32*(c&1)+32).
    EPI_RESIDUAL_SPLIT = 3,
};

struct ConvParams {
    const void* in;        // typed NHWC
    int in_cstride;        // elements between consecutive pixels of the input buffer
    long in_pstride;       // elements between consecutive 32-channel chunks (32 = interleaved NHWC; H*W*cstride =
                           // chunk-planar, the layout the RRDB trunk uses: every chunk read is a contiguous stream)
    int cin_chunks;        // number of 32-channel chunks to contract over
    int H, W;              // OUTPUT height/width (input is H/2 x W/2 when upsample2x)
    const void* wpk;       // packed weight fragments (see pack_conv3x3_weights)
    const float* bias;     // [32*cout_tiles] (zero padded)
    void* out;             // typed NHWC output (may be null for EPI_IMAGE)
    int out_cstride;       // elements between consecutive pixels of the output buffer
    long out_pstride;      // elements between the two 32-channel halves of a 64-channel output (32 = interleaved)
    int out_coff;          // first channel written
    float* out_f32;        // optional fp32 NHWC copy, stride = 32*cout_tiles (EPI_STORE/EPI_RESIDUAL)
    const float* res1;     // fp32 NHWC stride 32*cout_tiles
    const float* res2;     // optional second residual
    float s1, s2;
    uint8_t* out_u8;       // EPI_IMAGE: HxWx3 BGR uint8 (optional)
    uint16_t* out_u16;     // EPI_IMAGE: HxWx3 BGR uint16 = clamp -> x65535 -> rint (16-bit frames; optional)
    float* out_rgb;        // EPI_IMAGE: HxWx3 RGB float, un-clamped (optional)
    int img_H, img_W;      // EPI_IMAGE: size of the stored image (crop of the H x W conv output; mod-pad removal)
    int act;               // EPI_STORE: 1 = LeakyReLU(0.2), 2 = PReLU with per-output-channel slopes in chan_scale
    int upsample2x;        // 1 = input is nearest-neighbour x2 upsampled on the fly
    const float* chan_scale;  // EPI_RESIDUAL: optional per-output-channel factor, y = (acc+bias)*chan_scale[n]*s1 + res1
    int post_act;          // EPI_RESIDUAL: 1 = LeakyReLU(0.2) after the residual add (IFNet ResConv)
    int f32_cstride;       // NHWC fp32 side buffers: floats per pixel (0 = 32*cout_tiles) and first channel
    int f32_coff;
    int f32_native;        // 1: res1/res2/out_f32 use the accumulator-native layout (see f32_native_elems)
    // split trunk (EPI_RESIDUAL_SPLIT; out_lo also with 64-channel EPI_STORE): lo planes, strides of `out`
    void* out_lo;
    int n_id;              // residual planes (EPI_RESIDUAL_SPLIT), <= min(6, cin_chunks)
    float id_scale[6];     // exactly representable in the operand type
    float in_id_scale;     // EPI_RESIDUAL_SPLIT: conv chunks 0,1 (the conv's own input channels [0,64)) are also added as
                           // in_id_scale * I, from the tile already in LDS (0 = off)
    long chunk_off[6];     // EPI_RESIDUAL_SPLIT: byte offset of residual plane c from `in`; pixel stride = in_cstride
    unsigned long long* stamps;  // diagnostic builds only (-DFW_PAIR_STAMP)
    const void* zeros;     // >= 16 bytes of zeros in device memory (set by launch_conv3x3)
    // EPI_STORE / EPI_RESIDUAL: n_groups > 1 runs that many convolutions of 32 * cout_tiles output channels each in ONE launch
    // (grid.y = group): group g reads its weights wpk_gstride bytes further on and shifts bias, chan_scale, out_coff and f32_coff
    // by 32 * cout_tiles * g - the output-channel groups of a wide conv on a small feature map side by side (IFNet)
    // (with f32_native the groups' fp32 planes are f32_gstride floats apart: f32_native_elems(H, W, cout_tiles) or more)
    int n_groups;
    long wpk_gstride;
    long f32_gstride;
};

// Number of floats of an accumulator-native fp32 side buffer for an H x W problem with 32*cout_tiles channels:
// [tile][wave][row][ct][g][lane][4] over the padded 16x32 tile grid.
size_t f32_native_elems(int H, int W, int cout_tiles);

// Two chained 32-output-channel convs in one kernel (conv3x3_pair.hip): x_a = lrelu(conv_a(in[0:32*na])),
// x_b = lrelu(conv_b(cat(in[0:32*na], x_a))).
struct ConvPairParams {
    const void* in;        // operand-typed, chunk c / pixel (y,x) at c*in_pstride + (y*W + x)*in_cstride
    int in_cstride;
    long in_pstride;
    int na;                // 32-channel input chunks of conv_a (conv_b contracts na + 1 chunks)
    int H, W;
    const void* wpk_a;     // pack_conv3x3_weights(cout 32, cin 32*na)
    const float* bias_a;   // [32]
    const void* wpk_b;     // pack_conv3x3_weights(cout 32, cin 32*(na+1)); the last chunk multiplies x_a
    const float* bias_b;
    void* out_a;           // typed plane [H][W][out_cstride], 32 channels written
    void* out_b;
    int out_cstride;
    const void* zeros;     // set by launch_conv3x3_pair
    unsigned long long* stamps;  // diagnostic builds only (-DFW_PAIR_STAMP)
};
// conv5 of a dense block (64 output channels, the split-trunk residuals) as a row-wise Winograd F(2, 3) (conv3x3_wino.hip; f16 only):
// the ConvParams of launch_conv3x3(dt, 2, EPI_RESIDUAL_SPLIT, ...) with wpk = pack_conv3x3_wino_weights' fragments.
size_t pack_conv3x3_wino_weights(DType dt, const float* w, int cout, int cin, int cin_chunks, uint16_t* dst);
void launch_conv3x3_wino_split(const ConvParams& p, hipStream_t stream);
void launch_conv3x3_wino_store(const ConvParams& p, hipStream_t stream);   // act(conv + bias) -> typed planes (EPI_STORE, 64 channels)
void launch_conv3x3_pair(DType dt, const ConvPairParams& p, hipStream_t stream);
const void* conv_zero_page();  // 256 B of zeros on the current device

// lrelu(conv3x3(nearest_x2(x)) + bias), 64 -> 64 channels, as four 2x2 phase convolutions on the source grid (conv_up2x_phase.hip)
struct ConvUpParams {
    const void* in;        // operand-typed SOURCE image, chunk c / pixel (y,x) at c*in_pstride + (y*W + x)*in_cstride; 2 chunks
    int in_cstride;
    long in_pstride;
    int H, W;              // SOURCE size; the output is 2H x 2W
    const void* wpk;       // pack_conv_up2x_phase_weights
    const float* bias;     // [64]
    int act;               // 1 = LeakyReLU(0.2)
    void* out;             // typed [2H][2W][out_cstride]; the two 32-channel halves out_pstride elements apart (32 = interleaved)
    int out_cstride;
    long out_pstride;
    const void* zeros;     // set by the launcher
};
void launch_conv_up2x_phase(DType dt, const ConvUpParams& p, hipStream_t stream);
// w[64][64][3][3] fp32 -> the phase kernel's fragments (returns the number of uint16; dst may be null to query)
size_t pack_conv_up2x_phase_weights(DType dt, const float* w, uint16_t* dst);
int conv_num_cus();     // persistent workgroups per conv launch (CUs of the current device, FW_CONV_GRID)

// Launches the kernel; cout_tiles in {1,2} (32 or 64 output channels).
void launch_conv3x3(DType dt, int cout_tiles, ConvEpilogue epi, const ConvParams& p, hipStream_t stream);

// Host-side weight packer: torch layout w[cout][cin][3][3] fp32 -> MFMA A-fragments.
// Returns number of uint16 elements written (dst may be null to query).
size_t pack_conv3x3_weights(DType dt, const float* w, int cout, int cin, int cout_tiles, int cin_chunks,
                            uint16_t* dst);

// ---- frame <-> tensor conversions -----------------------------------------------------------
// uint8 BGR HxWx3 -> typed NHWC (32-channel padded), RGB order, /255.  unshuffle in {1,2}:
// 2 = pixel_unshuffle(2) front end of the x2 model (12 channels), with reflect mod-padding to even size.
void launch_u8_to_nhwc(DType dt, const uint8_t* in_bgr, int H, int W, void* out, int out_cstride,
                       int unshuffle, hipStream_t stream);
// the same for 8- or 16-bit samples (bits; 16: uint16 BGR, /65535)
void launch_frame_to_nhwc(DType dt, const void* in_bgr, int bits, int H, int W, void* out, int out_cstride, int unshuffle,
                          hipStream_t stream);
// SRVGGNetCompact tail: PixelShuffle(scale) of the last conv (fp32 [H][W][cstride], channel c*scale^2 + i*scale + j) plus
// the nearest-upsampled input, -> RGB float and/or clamp -> x255 -> rint -> uint8 BGR, both [scale*H][scale*W][3].
void launch_pixel_shuffle_add(const float* conv, int cstride, const uint8_t* in_bgr, int H, int W, int scale, uint8_t* out_bgr,
                              float* out_rgb, hipStream_t stream);
// the same for 8- or 16-bit samples (bits; 16: uint16 BGR base and output, /65535 and x65535)
void launch_pixel_shuffle_add_bits(const float* conv, int cstride, const void* in_bgr, int bits, int H, int W, int scale, void* out_bgr,
                                   float* out_rgb, hipStream_t stream);

// ---- NAFNet building blocks (nn_ops.hip) -------------------------------------------------------------
enum PointwiseMode : int {
    PW_STORE = 0,       // y = acc + bias -> typed and/or fp32 NHWC
    PW_RESIDUAL = 1,    // y = res + (acc + bias) * chan_scale -> fp32 NHWC
    PW_GATE = 2,        // SimpleGate fused: y[n] = (acc[n] + b[n]) * (acc[n + N/2] + b[n + N/2]) -> typed, N/2 channels
    PW_SHUFFLE_UP = 3,  // 1x1 conv + PixelShuffle(2) + skip add -> fp32 NHWC at twice the resolution
};

struct PointwiseParams {
    const void* a;         // [M][lda] operand-typed, or fp32 when a_f32
    int a_f32;
    long lda;              // elements per pixel row of a
    long M;                // output pixels
    int K;                 // contraction length (multiple of 32)
    int gather2x2;         // 1: 2x2 stride-2 conv, k = (dy*2+dx)*Cin + ci, a is the full-resolution [Hin][Win][lda] map
    int Win;               // gather2x2: input width; PW_SHUFFLE_UP: low-resolution width
    int Cin;               // gather2x2: input channels (K = 4*Cin)
    const float* a_scale;  // optional per-k scale applied while staging (SCA)
    const float* ln_w;     // LayerNorm2d fused into the staging (K == 64 == lda, a_f32, no gather): a = norm(a) * ln_w + ln_b
    const float* ln_b;     //   over the 64 channels of each pixel, eps = ln_eps; null = off
    float ln_eps;
    const void* wpk;       // pack_pointwise_weights
    const void* wpk16;     // optional: the same weights in pack_pointwise_weights16's layout - with it, typed inputs and
                           // K >= 128, cout % 256 == 0, the pipelined GEMM kernel (pointwise_gemm.hip) runs instead
    const float* bias;     // [32*N_tiles] or null
    int N_tiles;           // cout / 32
    int mode;
    void* out_typed;
    long ldo;
    float* out_f32;
    long ldf;
    const float* res_f32;
    const float* chan_scale;
};

void launch_pointwise(DType dt, const PointwiseParams& p, hipStream_t st);
size_t pack_pointwise_weights(DType dt, const float* w, int cout, int K, uint16_t* dst);
// the second half of a width-64 NAFBlock (conv3 + beta residual + LayerNorm2d + conv4 + SimpleGate + conv5 + gamma residual) in
// one pass over the fp32 stream; weights in pack_pointwise_weights' layout
void launch_naf_tail64(DType dt, const void* x, const float* a_scale, float* stream, long M, const void* w3, const float* b3,
                       const float* beta, const float* ln_w, const float* ln_b, float ln_eps, const void* w4, const float* b4,
                       const void* w5, const float* b5, const float* gamma, hipStream_t st);
// naf_tail128.hip: the same second half at 128 channels, the three weight matrices streamed through LDS in eight blocks
struct NafTail128Params {
    const void* x;          // typed [M][ldx]: the gated tensor of the block's first half (128 channels)
    long ldx;
    float* stream;          // fp32 [M][lds_], updated in place
    long lds_;
    long M;
    float ln_eps;
    const void* blocks;     // pack_naf_tail128_blocks
    void* w3_scratch;       // 2 blocks of scratch: conv3 with this forward's SCA factors folded in (written by the launcher)
    const void* w3_scaled;  // set by the launcher
};
size_t naf_tail128_block_bytes();
void launch_naf_tail128(DType dt, const NafTail128Params& p, const float* sca, hipStream_t st);
void pack_naf_tail128_blocks(DType dt, const float* w3, const float* b3, const float* beta, const float* ln_w, const float* ln_b, const float* w4,
                             const float* b4, const float* w5, const float* b5, const float* gamma, void* dst);
// pointwise_gemm.hip: the many-channel form (256 x 256 tiles, LDS-DMA pipeline)
bool pointwise_gemm_eligible(const PointwiseParams& p);
void launch_pointwise_gemm(DType dt, const PointwiseParams& p, hipStream_t st);
size_t pack_pointwise_weights16(DType dt, const float* w, int cout, int K, int gate, uint16_t* dst);
// dst = src (pack_pointwise_weights16 layout, cout N x K) with input channel k scaled by scale[k] (NAFNet's SCA in front of conv3)
void launch_pw16_scale_weights(DType dt, const void* src, const float* scale, int N, int K, void* dst, hipStream_t st);
// pw_dw_fused.hip: LayerNorm + 1x1 conv + depthwise 3x3 [+ gate] in one kernel (NAFBlock front at c = 64 / 128, Restormer's
// qkv and GDFN fronts at c = 48 / 96)
enum PwDwMode : int {
    PWDW_NONE = 0,       // out = dwconv(conv(norm(x))), all N channels
    PWDW_GATE_MUL = 1,   // out[i] = dw[i] * dw[N/2 + i] (SimpleGate) + pooled sums of the output
    PWDW_GATE_GELU = 2,  // out[i] = gelu(dw[i]) * dw[N/2 + i] (GDFN gate, exact erf GELU)
};
struct PwDwParams {
    const float* x;        // fp32 NHWC stream, ldx floats per pixel, cin real channels
    long ldx;
    int H, W, cin;
    float ln_eps;          // LayerNorm over the cin channels; its affine part is folded into the parameter blocks
    const void* blocks;    // pack_pw_dw_blocks
    int n_chunks;          // N / 64
    int mode;
    void* out;             // typed NHWC, ldo elements per pixel: N channels (PWDW_NONE) or N / 2 (gate modes)
    long ldo;
    float* partial;        // PWDW_GATE_MUL: [pw_dw_blocks][N / 2] sums of the gated output (SCA pooling), or null
    // PWDW_NONE, Restormer's qkv: with qT set, chunks [0, t_chunks) (q, then k: channel rows 64 j of chunk j) are written ONLY
    // there, as [pixel group of 8][t_ld channels][8 pixels] - the operand layout of the Gram kernel (attn_gram_mfma_kernel) - with
    // the pixels in tile order: tile t owns groups [56 t, 56 t + 56) (14 x 30 pixels row-major, zero padded to 448); the chunks
    // behind them (v) go to `out` from its channel 0 on.  q and k need not be whole chunks each (96 + 96 channels = 3 chunks).
    void* qT;
    int t_chunks;
    long t_ld;
};
long pw_dw_transposed_pixels(int H, int W);   // pixels of qT including the padding (a multiple of 32)
void launch_attn_matrix_from_transposed(DType dt, const void* qT, const void* kT, long Mp, int ldc, int heads, int ch, const float* temperature,
                                        float* workspace, float* attn, hipStream_t st);
bool pw_dw_eligible(int cin, int mode);
int pw_dw_blocks(int H, int W);   // grid of the kernel = rows of `partial`
void launch_pw_dw(DType dt, const PwDwParams& p, hipStream_t st);
size_t pack_pw_dw_blocks(DType dt, const float* w, const float* bias, const float* ln_w, const float* ln_b, const float* wdw, const float* bdw,
                         int N, int c, int gate, void* dst);
void launch_layernorm2d(DType dt, const float* x, long M, int C, const float* w, const float* b, void* out,
                        hipStream_t st);
int dwconv_blocks(int H, int W, int C);  // grid of the dwconv kernel = rows of its `partial` output
void launch_dwconv3x3_gate(DType dt, const void* x, int H, int W, int C, const float* wdw, const float* bdw, void* out,
                           float* partial /* [dwconv_blocks][C] or null */, hipStream_t st);
void launch_sca(const float* partial, int nblocks, long HW, int C, const float* w, const float* b, float* s,
                hipStream_t st);
void launch_f32_to_planar(DType dt, const float* x, long M, int C, void* out, hipStream_t st);

// AESRGAN (aesrgan.hip): fp32 RGB [M][3] -> typed [M][32] (channels 3.. zero); fp32 [M][cstride] -> fp32 RGB [M][3]
void launch_rgb_f32_to_nhwc(DType dt, const float* x, long M, void* out, hipStream_t st);
void launch_take_rgb_f32(const float* src, int cstride, long M, float* out, hipStream_t st);

// ---- TAP frame path + K8 blend kernels (frame_ops.hip) ---------------------------------------------------
void launch_u8_to_nhwc_padded(DType dt, const uint8_t* in_bgr, int H, int W, int Hp, int Wp, void* out, hipStream_t st);
void launch_tap_post(const uint8_t* in_bgr, const float* rgb, int H, int W, int Wp, int rgb_cstride, uint8_t* out_bgr,
                     float* out_rgb, hipStream_t st);
void launch_u8_crop(const uint8_t* src, int W, int y0, int x0, int th, int tw, uint8_t* dst, hipStream_t st);
void launch_tile_blend_acc(float* acc, float* wsum, int W, const uint8_t* tile, int y0, int x0, int th, int tw, int ov,
                           int top, int bottom, int left, int right, hipStream_t st);
void launch_tile_blend_finish(const float* acc, const float* wsum, long npix, uint8_t* out, hipStream_t st);
void launch_temporal_average(const uint8_t* const* frames, const float* weights, int count, long n, uint8_t* out,
                             hipStream_t st);
void launch_strength_blend(const uint8_t* orig, const uint8_t* den, float one_minus_s, float s, long n, uint8_t* out,
                           hipStream_t st);

void launch_flow_accumulate(const uint8_t* frame, const float* fx, const float* fy, const float* wmap, double wscale,
                            const float* mag, float thr, int inverse, int H, int W, double* acc, double* wsum, hipStream_t st);
// grain add-back of the TAP driver (tap_denoise.py:621-632); tmp: H*W uint16 scratch
void launch_grain_addback(const uint8_t* orig, const uint8_t* den, int H, int W, double factor, uint16_t* tmp, uint8_t* out,
                          hipStream_t st);
// cv2.resize(INTER_LANCZOS4) on 8-bit H x W x C images (device pointers); synchronises the stream
void launch_resize_lanczos4_u8(const uint8_t* src, int Hs, int Ws, int C, uint8_t* dst, int Hd, int Wd, hipStream_t st);
void launch_resize_linear_u8(const uint8_t* src, int Hs, int Ws, int C, uint8_t* dst, int Hd, int Wd, hipStream_t st);
void launch_face_paste_u8(uint8_t* frame, int H, int W, int x1, int y1, int x2, int y2, const uint8_t* enh, float strength, hipStream_t st);
// the same on 16-bit images (OpenCV's float path for ushort); synchronises the stream
void launch_resize_lanczos4_u16(const uint16_t* src, int Hs, int Ws, int C, uint16_t* dst, int Hd, int Wd, hipStream_t st);
void launch_flow_accumulate_finish(const double* acc, const double* wsum, long n, uint8_t* out, hipStream_t st);
// `_preserve_edges` (temporal_denoise.py:1636-1667): Canny edge mask of `orig` blends it over `den`; synchronises the stream
size_t preserve_edges_scratch_bytes(int H, int W);
void launch_preserve_edges(const uint8_t* orig, const uint8_t* den, int H, int W, int lo, int hi, void* scratch, uint8_t* out,
                           hipStream_t st);

// ---- IFNet building blocks (ifnet_ops.hip), used by the whole-model engine in ifnet.hip -------------------------------------
void launch_ifnet_u8_to_rgb(const uint8_t* in_bgr, int H, int W, int Hp, int Wp, float* out, hipStream_t st);
void launch_resize_bilinear(const float* src, int Hs, int Ws, int C, float* dst, int Hd, int Wd, int dst_cstride, int dst_coff,
                            float scale_factor, float mul, hipStream_t st);
void launch_ifnet_build_x(const float* i0, const float* i1, const float* flow, const float* mask, int H, int W, float timestep,
                          float* x, hipStream_t st);
void launch_unshuffle2_cast(DType dt, const void* src, bool src_f32, int h, int w, int C, int src_cstride, void* dst,
                            int dst_channels, hipStream_t st);
void launch_depth_to_space4(const float* src, int h, int w, int cs, float* dst, hipStream_t st);
// build_x + both resizes + cat + pixel_unshuffle(2) + cast of an IFBlock's input in one kernel; lastconv's depth-to-space inside the accumulate
void launch_ifnet_stage_input(DType dt, const float* i0, const float* i1, const float* flow, const float* mask, int H, int W, float timestep,
                              int s, void* dst, int dst_channels, hipStream_t st);
void launch_ifnet_accumulate_d2s(const float* t96, int hf, int wf, int cs, int H, int W, float scale, float* flow, float* mask, int first, hipStream_t st);
void launch_ifnet_accumulate(const float* tmp, int hs, int ws, int H, int W, float scale, float* flow, float* mask, int first,
                             hipStream_t st);
void launch_ifnet_blend(const float* i0, const float* i1, const float* flow, const float* mask, int Hp, int Wp, int H, int W,
                        uint8_t* out_bgr, float* out_rgb, hipStream_t st);

// thread-local message returned by fw_last_error()
std::string& last_error_ref();

uint16_t f32_to_operand(DType dt, float f);
float operand_to_f32(DType dt, uint16_t v);

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define FW_HIP_CHECK(expr)                                                                        \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            (void)hipGetLastError(); /* clear the sticky error so the handle stays usable */      \
            int _code = (_e == hipErrorOutOfMemory) ? 2 : 3;                                      \
            throw fw::Error(_code, std::string(_e == hipErrorOutOfMemory ? "GPU out of memory: " : \
                                                                           "HIP error: ") +       \
                                       hipGetErrorString(_e) + " at " + __FILE__ + ":" +          \
                                       std::to_string(__LINE__));                                 \
        }                                                                                         \
    } while (0)

// Makes `dev` the calling thread's current device for the scope of a C-ABI entry (one definition for every engine).
struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) {
        FW_HIP_CHECK(hipGetDevice(&prev));
        if (prev != dev) FW_HIP_CHECK(hipSetDevice(dev));
        else prev = -1;
    }
    ~DevGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
};

// One engine handle = one workspace arena, but its callers may enqueue on different streams (thread-pool callers with a stream
// each, include/framewright_hip.h).  The handle's mutex serialises the ENQUEUE only; this orders the device work: every forward
// records an event behind its last launch, and a forward enqueued on another stream first waits for that event - the workspace is
// never written by two forwards at once.  Same stream: stream order already does it, no wait is inserted.
struct StreamOrder {
    hipEvent_t ev = nullptr;
    hipStream_t last = nullptr;
    bool have = false;
    struct Scope {
        StreamOrder& o;
        hipStream_t st;
        Scope(StreamOrder& order, hipStream_t stream) : o(order), st(stream) {
            if (o.have && o.last != st) FW_HIP_CHECK(hipStreamWaitEvent(st, o.ev, 0));
        }
        ~Scope() {
            if (!o.ev && hipEventCreateWithFlags(&o.ev, hipEventDisableTiming) != hipSuccess) {
                o.ev = nullptr;
                (void)hipGetLastError();
                (void)hipStreamSynchronize(st);   // no event to be had: fall back to a host wait, never to an unordered workspace
                o.have = false;
                return;
            }
            if (hipEventRecord(o.ev, st) == hipSuccess) {
                o.last = st;
                o.have = true;
            } else {
                (void)hipGetLastError();
                (void)hipStreamSynchronize(st);
                o.have = false;
            }
        }
        Scope(const Scope&) = delete;
        Scope& operator=(const Scope&) = delete;
    };
    void destroy() {
        if (ev) (void)hipEventDestroy(ev);
        ev = nullptr;
        have = false;
    }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
// GELU (exact form, 0.5 x (1 + erf(x / sqrt 2))) of two values, erf after Abramowitz & Stegun 7.1.28:
//   erf(z) = 1 - 1 / (1 + a1 z + ... + a6 z^6)^16,  z >= 0, |error| <= 3e-7 (7e-7 on the GELU in fp32, against scipy on [-12, 12])
// - one reciprocal, no exponential: Horner and the four squarings run as packed fp32 (v_pk_fma_f32 / v_pk_mul_f32), ~11 issue
// slots per value where 7.1.26 (a reciprocal AND an exponential, scalar) took ~21.  The depthwise phase of the GDFN kernel is
// VALU-bound and the gate was 60 % of it.  Large |x|: the power overflows to inf, 1 / inf = 0, erf = 1.
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 ax = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
    const f32x2 z = ax * 0.70710678118654752f;
    f32x2 q = z * 0.0000430638f + 0.0002765672f;
    q = q * z + 0.0001520143f;
    q = q * z + 0.0092705272f;
    q = q * z + 0.0422820123f;
    q = q * z + 0.0705230784f;
    q = q * z + 1.0f;
    q = q * q;
    q = q * q;
    q = q * q;
    q = q * q;
    const f32x2 r = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
    const f32x2 h = ax * 0.5f;          // 0.5 (x + |x| erf) = (0.5 x + 0.5 |x|) - 0.5 |x| r
    const f32x2 s = x * 0.5f + h;
    return s - h * r;
}
// the same, one value, scalar instructions (where register pairs for the packed form are not to be had)
__device__ __forceinline__ float gelu_erf1(float x) {
    const float ax = __builtin_fabsf(x);
    const float z = ax * 0.70710678118654752f;
    float q = __builtin_fmaf(z, 0.0000430638f, 0.0002765672f);
    q = __builtin_fmaf(q, z, 0.0001520143f);
    q = __builtin_fmaf(q, z, 0.0092705272f);
    q = __builtin_fmaf(q, z, 0.0422820123f);
    q = __builtin_fmaf(q, z, 0.0705230784f);
    q = __builtin_fmaf(q, z, 1.0f);
    q = q * q;
    q = q * q;
    q = q * q;
    q = q * q;
    const float r = __builtin_amdgcn_rcpf(q);
    const float hh = ax * 0.5f;
    return __builtin_fmaf(-hh, r, __builtin_fmaf(x, 0.5f, hh));
}

}  // namespace fw
