// Frame <-> tensor conversion kernels (K5 of SURVEY.md §8a).  HBM-bound byte work: one thread per output
// pixel, 16-byte vector stores, no LDS.
//
// Follows the pre-processing of RealESRGANer.enhance / pre_process as recorded in SURVEY.md §A.2 and
// called from reference src/framewright/processors/pytorch_realesrgan.py:223: BGR->RGB, /255, HWC->CHW
// (here NHWC), for the x2 model reflect mod-pad to an even size followed by pixel_unshuffle(2)
// (basicsr RRDBNet.forward).
#include <cmath>
#include <cstring>
#include <vector>
#include "fw_internal.h"

// The blend kernels below restate float32 numpy arithmetic bit for bit: a*b+c must round twice, so this translation
// unit is compiled with -ffp-contract=off (build.py PER_FILE_FLAGS; hipcc's default is fast).

namespace fw {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T>
__device__ __forceinline__ uint4 pack8(const float* v);
template <>
__device__ __forceinline__ uint4 pack8<__bf16>(const float* v) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
    return __builtin_bit_cast(uint4, o);
}
template <>
__device__ __forceinline__ uint4 pack8<_Float16>(const float* v) {
    f16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (_Float16)v[i];
    return __builtin_bit_cast(uint4, o);
}

// reflect index (torch 'reflect' padding on the right/bottom edge): n + k -> n - 2 - k
__device__ __forceinline__ int reflect_hi(int i, int n) { return i < n ? i : 2 * n - 2 - i; }

// UNSHUFFLE == 1: out[y][x][c] = in[y][x][2 - c] / 255 for c < 3, zero for 3 <= c < 32.
// UNSHUFFLE == 2: out is ceil(H/2) x ceil(W/2); channel c*4 + dy*2 + dx = RGB channel c of input pixel
//                 (2y+dy, 2x+dx) (torch.pixel_unshuffle order), zero for 12 <= ch < 32.
// IN = uint8_t (/255) or uint16_t (/65535: RealESRGANer.enhance treats an image whose maximum exceeds 256 as 16-bit).
template <typename T, int UNSHUFFLE, typename IN>
__global__ __launch_bounds__(256) void u8_to_nhwc_kernel(const IN* __restrict__ in, int H, int W, T* out,
                                                         int out_cstride, int Ho, int Wo) {
    constexpr float INV = sizeof(IN) == 1 ? 1.f / 255.f : 1.f / 65535.f;
    const long n = (long)Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / Wo);
        const int x = (int)(i - (long)y * Wo);
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.f;
        if constexpr (UNSHUFFLE == 1) {
            const IN* px = in + ((size_t)y * W + x) * 3;
            v[0] = px[2] * INV;
            v[1] = px[1] * INV;
            v[2] = px[0] * INV;
        } else {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int sy = reflect_hi(2 * y + dy, H);
                    const int sx = reflect_hi(2 * x + dx, W);
                    const IN* px = in + ((size_t)sy * W + sx) * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c * 4 + dy * 2 + dx] = px[2 - c] * INV;
                }
        }
        uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * out_cstride);
        o[0] = pack8<T>(v);
        o[1] = pack8<T>(v + 8);
        o[2] = make_uint4(0, 0, 0, 0);
        o[3] = make_uint4(0, 0, 0, 0);
    }
}

void launch_frame_to_nhwc(DType dt, const void* in_bgr, int bits, int H, int W, void* out, int out_cstride, int unshuffle,
                          hipStream_t stream) {
    if (out_cstride < 32 || (out_cstride & 7)) throw Error(1, "u8_to_nhwc: bad channel stride");
    if (bits != 8 && bits != 16) throw Error(1, "frame_to_nhwc: 8- or 16-bit samples expected");
    const int Ho = unshuffle == 2 ? (H + 1) / 2 : H;
    const int Wo = unshuffle == 2 ? (W + 1) / 2 : W;
    if (unshuffle == 2 && (H < 2 || W < 2)) throw Error(1, "u8_to_nhwc: x2 model needs at least 2x2 input");
    const long n = (long)Ho * Wo;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    dim3 grid(blocks), block(256);
#define FW_LAUNCH(T, U, IN) \
    hipLaunchKernelGGL((u8_to_nhwc_kernel<T, U, IN>), grid, block, 0, stream, (const IN*)in_bgr, H, W, (T*)out, out_cstride, Ho, Wo)
#define FW_LAUNCH_T(T)                                                            \
    do {                                                                          \
        if (bits == 8) {                                                          \
            if (unshuffle == 2) FW_LAUNCH(T, 2, uint8_t); else FW_LAUNCH(T, 1, uint8_t);    \
        } else {                                                                  \
            if (unshuffle == 2) FW_LAUNCH(T, 2, uint16_t); else FW_LAUNCH(T, 1, uint16_t);  \
        }                                                                         \
    } while (0)
    if (dt == DT_BF16) FW_LAUNCH_T(__bf16); else FW_LAUNCH_T(_Float16);
#undef FW_LAUNCH_T
#undef FW_LAUNCH
    FW_HIP_CHECK(hipGetLastError());
}

void launch_u8_to_nhwc(DType dt, const uint8_t* in_bgr, int H, int W, void* out, int out_cstride, int unshuffle,
                       hipStream_t stream) {
    launch_frame_to_nhwc(dt, in_bgr, 8, H, W, out, out_cstride, unshuffle, stream);
}


// ---- classical motion-compensated temporal denoise (reference src/framewright/processors/temporal_denoise.py:440-477
//      `warp_frame` and :1521-1580 `_denoise_with_flow`): per neighbour frame  aligned = cv2.remap(frame, x + flow_x,
//      y + flow_y, INTER_LINEAR, BORDER_REFLECT_101);  weight = exp(-d*decay) * confidence, halved where the flow magnitude
//      exceeds its 90th percentile;  accumulated += aligned (float64) * weight;  weight_sum += weight;  result =
//      (accumulated / max(weight_sum, 1e-6)).astype(uint8).
//      The remap restates OpenCV's 8-bit INTER_LINEAR arithmetic (coordinates rounded to 1/32 pixel with round-half-even,
//      15-bit coefficients, (sum + 2^14) >> 15); cv2 is absent from the build container, so that restatement is unpinned.
//      The dense flow itself (cv2 Farneback / DIS) stays on the host: it is not part of this path. --------------------------
__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - p - 2;
    return p;
}

__global__ __launch_bounds__(256) void flow_accumulate_kernel(const uint8_t* __restrict__ frame, const float* __restrict__ fx,
                                                              const float* __restrict__ fy, const float* __restrict__ wmap,
                                                              double wscale, const float* __restrict__ mag, float thr,
                                                              int inverse, int H, int W, double* acc, double* wsum) {
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        double v[3];
        if (fx) {
            // map = (grid +/- flow).astype(float32); cv2.remap fixed-point: round(map * 32), 5 fractional bits
            const float mx = inverse ? (float)((double)x - (double)fx[i]) : (float)((double)x + (double)fx[i]);
            const float my = inverse ? (float)((double)y - (double)fy[i]) : (float)((double)y + (double)fy[i]);
            const int sx = (int)rintf(mx * 32.0f), sy = (int)rintf(my * 32.0f);
            const int ix = sx >> 5, iy = sy >> 5, ax = sx & 31, ay = sy & 31;
            const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
            const int x0 = reflect101(ix, W), x1 = reflect101(ix + 1, W), y0 = reflect101(iy, H), y1 = reflect101(iy + 1, H);
            const uint8_t* p00 = frame + ((size_t)y0 * W + x0) * 3;
            const uint8_t* p01 = frame + ((size_t)y0 * W + x1) * 3;
            const uint8_t* p10 = frame + ((size_t)y1 * W + x0) * 3;
            const uint8_t* p11 = frame + ((size_t)y1 * W + x1) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v[c] = (double)((p00[c] * w00 + p01[c] * w01 + p10[c] * w10 + p11[c] * w11 + (1 << 14)) >> 15);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (double)frame[(size_t)i * 3 + c];
        }
        double w = wmap ? wscale * (double)wmap[i] : wscale;
        if (mag && mag[i] > thr) w *= 0.5;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[(size_t)i * 3 + c] += v[c] * w;
        wsum[i] += w;
    }
}

__global__ __launch_bounds__(256) void flow_accumulate_finish_kernel(const double* __restrict__ acc, const double* __restrict__ wsum,
                                                                     long n, uint8_t* out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double d = fmax(wsum[i], 1e-6);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(size_t)i * 3 + c] = (uint8_t)(acc[(size_t)i * 3 + c] / d);
    }
}

void launch_flow_accumulate(const uint8_t* frame, const float* fx, const float* fy, const float* wmap, double wscale,
                            const float* mag, float thr, int inverse, int H, int W, double* acc, double* wsum, hipStream_t st) {
    const long n = (long)H * W;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(flow_accumulate_kernel, dim3(blocks), dim3(256), 0, st, frame, fx, fy, wmap, wscale, mag, thr, inverse, H, W,
                       acc, wsum);
    FW_HIP_CHECK(hipGetLastError());
}

void launch_flow_accumulate_finish(const double* acc, const double* wsum, long n, uint8_t* out, hipStream_t st) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(flow_accumulate_finish_kernel, dim3(blocks), dim3(256), 0, st, acc, wsum, n, out);
    FW_HIP_CHECK(hipGetLastError());
}

// ---- SRVGGNetCompact tail (Real-ESRGAN realesr-animevideov3 / realesr-general-x4v3; reference model table
//      src/framewright/processors/pytorch_realesrgan.py:119-128 declares them, the network itself is third-party) ---------
// S = uint8_t (range 255) or uint16_t (range 65535: RealESRGANer.enhance normalises a 16-bit image by 65535 and returns uint16)
template <typename S>
__global__ __launch_bounds__(256) void pixel_shuffle_add_kernel(const float* __restrict__ conv, int cstride,
                                                                const S* __restrict__ in_bgr, int H, int W, int s,
                                                                S* out_bgr, float* out_rgb) {
    constexpr float TOP = sizeof(S) == 1 ? 255.0f : 65535.0f;
    const int Wo = W * s;
    const long n = (long)H * s * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int Y = (int)(i / Wo), X = (int)(i - (long)Y * Wo);
        const int y = Y / s, x = X / s, sub = (Y - y * s) * s + (X - x * s);
        const float* c = conv + ((size_t)y * W + x) * cstride;
        const S* px = in_bgr + ((size_t)y * W + x) * 3;
        float v[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) v[ch] = c[ch * s * s + sub] + px[2 - ch] / TOP;
        if (out_rgb) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) out_rgb[(size_t)i * 3 + ch] = v[ch];
        }
        if (out_bgr) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
                out_bgr[(size_t)i * 3 + 2 - ch] = (S)rintf(fminf(fmaxf(v[ch], 0.f), 1.f) * TOP);
        }
    }
}

void launch_pixel_shuffle_add_bits(const float* conv, int cstride, const void* in_bgr, int bits, int H, int W, int scale, void* out_bgr,
                                   float* out_rgb, hipStream_t st) {
    if (scale < 1 || scale > 4 || cstride < 3 * scale * scale) throw Error(1, "pixel_shuffle_add: bad scale / channel stride");
    if (bits != 8 && bits != 16) throw Error(1, "pixel_shuffle_add: 8- or 16-bit samples expected");
    const long n = (long)H * scale * W * scale;
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    if (bits == 8)
        hipLaunchKernelGGL((pixel_shuffle_add_kernel<uint8_t>), dim3(blocks), dim3(256), 0, st, conv, cstride, (const uint8_t*)in_bgr, H, W, scale,
                           (uint8_t*)out_bgr, out_rgb);
    else
        hipLaunchKernelGGL((pixel_shuffle_add_kernel<uint16_t>), dim3(blocks), dim3(256), 0, st, conv, cstride, (const uint16_t*)in_bgr, H, W, scale,
                           (uint16_t*)out_bgr, out_rgb);
    FW_HIP_CHECK(hipGetLastError());
}

void launch_pixel_shuffle_add(const float* conv, int cstride, const uint8_t* in_bgr, int H, int W, int scale, uint8_t* out_bgr,
                              float* out_rgb, hipStream_t st) {
    launch_pixel_shuffle_add_bits(conv, cstride, in_bgr, 8, H, W, scale, out_bgr, out_rgb, st);
}

// ---- AESRGAN (csrc/aesrgan.hip): fp32 RGB [M][3] <-> the conv kernels' layouts ----------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rgb_f32_to_nhwc_kernel(const float* __restrict__ x, long M, T* out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long)gridDim.x * blockDim.x) {
        float v[8] = {x[i * 3], x[i * 3 + 1], x[i * 3 + 2], 0, 0, 0, 0, 0};
        uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 32);
        o[0] = pack8<T>(v);
        o[1] = make_uint4(0, 0, 0, 0);
        o[2] = make_uint4(0, 0, 0, 0);
        o[3] = make_uint4(0, 0, 0, 0);
    }
}
void launch_rgb_f32_to_nhwc(DType dt, const float* x, long M, void* out, hipStream_t st) {
    const int blocks = (int)((M + 255) / 256 < 2048 ? (M + 255) / 256 : 2048);
    if (dt == DT_BF16)
        hipLaunchKernelGGL((rgb_f32_to_nhwc_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, x, M, (__bf16*)out);
    else
        hipLaunchKernelGGL((rgb_f32_to_nhwc_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, x, M, (_Float16*)out);
    FW_HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void take_rgb_f32_kernel(const float* __restrict__ src, int cstride, long M, float* out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M * 3; i += (long)gridDim.x * blockDim.x) {
        const long px = i / 3;
        out[i] = src[px * cstride + (i - px * 3)];
    }
}
void launch_take_rgb_f32(const float* src, int cstride, long M, float* out, hipStream_t st) {
    const long n = M * 3;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(take_rgb_f32_kernel, dim3(blocks), dim3(256), 0, st, src, cstride, M, out);
    FW_HIP_CHECK(hipGetLastError());
}

// ---- TAP (NAFNet) frame path ---------------------------------------------------------------------------------
// uint8 BGR H x W x 3 -> typed [Hp][Wp][32] RGB/255 with zeros outside H x W (NAFNet.check_image_size zero pad,
// SURVEY.md §A.3; pre-processing reference src/framewright/processors/tap_denoise.py:373-397).
template <typename T>
__global__ __launch_bounds__(256) void u8_to_nhwc_padded_kernel(const uint8_t* __restrict__ in, int H, int W, int Hp, int Wp,
                                                                T* out) {
    const long n = (long)Hp * Wp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / Wp), x = (int)(i - (long)y * Wp);
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (y < H && x < W) {
            const uint8_t* px = in + ((size_t)y * W + x) * 3;
            v[0] = px[2] / 255.0f;  // the reference divides (frame.astype(float32) / 255.0), it does not multiply by 1/255
            v[1] = px[1] / 255.0f;
            v[2] = px[0] / 255.0f;
        }
        uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 32);
        o[0] = pack8<T>(v);
        o[1] = make_uint4(0, 0, 0, 0);
        o[2] = make_uint4(0, 0, 0, 0);
        o[3] = make_uint4(0, 0, 0, 0);
    }
}

void launch_u8_to_nhwc_padded(DType dt, const uint8_t* in_bgr, int H, int W, int Hp, int Wp, void* out, hipStream_t st) {
    const long n = (long)Hp * Wp;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (dt == DT_BF16)
        hipLaunchKernelGGL((u8_to_nhwc_padded_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, in_bgr, H, W, Hp, Wp, (__bf16*)out);
    else
        hipLaunchKernelGGL((u8_to_nhwc_padded_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, in_bgr, H, W, Hp, Wp,
                           (_Float16*)out);
    FW_HIP_CHECK(hipGetLastError());
}

// out = ending(x) + inp, crop to H x W, then tap_denoise.py:399-415: np.clip(x * 255.0, 0, 255).astype(uint8)
// (TRUNCATION, not rounding), RGB -> BGR.
__global__ __launch_bounds__(256) void tap_post_kernel(const uint8_t* __restrict__ in_bgr, const float* __restrict__ rgb,
                                                       int H, int W, int Wp, int cs, uint8_t* out_bgr, float* out_rgb) {
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        const uint8_t* px = in_bgr + (size_t)i * 3;
        const float* r = rgb + ((size_t)y * Wp + x) * cs;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = r[c] + px[2 - c] / 255.0f;
        if (out_rgb) {
#pragma unroll
            for (int c = 0; c < 3; ++c) out_rgb[(size_t)i * 3 + c] = v[c];
        }
        if (out_bgr) {
#pragma unroll
            for (int c = 0; c < 3; ++c) out_bgr[(size_t)i * 3 + (2 - c)] = (uint8_t)fminf(fmaxf(v[c] * 255.0f, 0.f), 255.f);
        }
    }
}

void launch_tap_post(const uint8_t* in_bgr, const float* rgb, int H, int W, int Wp, int cs, uint8_t* out_bgr, float* out_rgb,
                     hipStream_t st) {
    const long n = (long)H * W;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(tap_post_kernel, dim3(blocks), dim3(256), 0, st, in_bgr, rgb, H, W, Wp, cs, out_bgr, out_rgb);
    FW_HIP_CHECK(hipGetLastError());
}

// ---- K8: tile ramp blend, temporal weighted average, strength blend (uint8 truncation semantics) ------------------
__global__ __launch_bounds__(256) void u8_crop_kernel(const uint8_t* __restrict__ src, int W, int y0, int x0, int th, int tw,
                                                      uint8_t* dst) {
    const long n = (long)th * tw * 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / (tw * 3)), rem = (int)(i - (long)y * tw * 3);
        dst[i] = src[((size_t)(y0 + y) * W + x0) * 3 + rem];
    }
}

void launch_u8_crop(const uint8_t* src, int W, int y0, int x0, int th, int tw, uint8_t* dst, hipStream_t st) {
    const long n = (long)th * tw * 3;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(u8_crop_kernel, dim3(blocks), dim3(256), 0, st, src, W, y0, x0, th, tw, dst);
    FW_HIP_CHECK(hipGetLastError());
}

// numpy.linspace(0, 1, n)[k] in float64: k * (1/(n-1)), last element exactly 1; linspace(0, 1, 1) == [0.]
__device__ __forceinline__ double linspace01(int k, int n) {
    if (n == 1) return 0.0;
    return (k == n - 1) ? 1.0 : (double)k * (1.0 / (double)(n - 1));
}

// tap_denoise.py:461-486: tile_weight = ones; *= ramp on interior edges (float32 array *= float64 ramp, in the order
// top, bottom, left, right); output += tile_result * tile_weight; weight += tile_weight (all float32)
__global__ __launch_bounds__(256) void tile_blend_acc_kernel(float* acc, float* wsum, int W, const uint8_t* __restrict__ tile,
                                                             int y0, int x0, int th, int tw, int ov, int top, int bottom,
                                                             int left, int right) {
    const long n = (long)th * tw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / tw), x = (int)(i - (long)y * tw);
        float wgt = 1.0f;
        if (ov > 0) {
            if (top && y < ov) wgt = (float)((double)wgt * linspace01(y, ov));
            if (bottom && y >= th - ov) wgt = (float)((double)wgt * linspace01(th - 1 - y, ov));
            if (left && x < ov) wgt = (float)((double)wgt * linspace01(x, ov));
            if (right && x >= tw - ov) wgt = (float)((double)wgt * linspace01(tw - 1 - x, ov));
        }
        const size_t o = (size_t)(y0 + y) * W + (x0 + x);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[o * 3 + c] = __fadd_rn(acc[o * 3 + c], __fmul_rn((float)tile[(size_t)i * 3 + c], wgt));
        wsum[o] = __fadd_rn(wsum[o], wgt);
    }
}

void launch_tile_blend_acc(float* acc, float* wsum, int W, const uint8_t* tile, int y0, int x0, int th, int tw, int ov,
                           int top, int bottom, int left, int right, hipStream_t st) {
    const long n = (long)th * tw;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(tile_blend_acc_kernel, dim3(blocks), dim3(256), 0, st, acc, wsum, W, tile, y0, x0, th, tw, ov, top,
                       bottom, left, right);
    FW_HIP_CHECK(hipGetLastError());
}

// output = (output / max(weight, 1e-8)).astype(uint8)
__global__ __launch_bounds__(256) void tile_blend_finish_kernel(const float* __restrict__ acc, const float* __restrict__ wsum,
                                                                long npix, uint8_t* out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const float w = fmaxf(wsum[i], 1e-8f);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[i * 3 + c] = (uint8_t)(int)__fdiv_rn(acc[i * 3 + c], w);
    }
}

void launch_tile_blend_finish(const float* acc, const float* wsum, long npix, uint8_t* out, hipStream_t st) {
    const int blocks = (int)((npix + 255) / 256 < 2048 ? (npix + 255) / 256 : 2048);
    hipLaunchKernelGGL(tile_blend_finish_kernel, dim3(blocks), dim3(256), 0, st, acc, wsum, npix, out);
    FW_HIP_CHECK(hipGetLastError());
}

// tap_denoise.py:521-534: result = zeros(float32); result += frame_k(float32) * w_k (w_k rounded to float32, product
// and sum each rounded to float32, no fma); astype(uint8)
struct TemporalArgs {
    const uint8_t* frames[16];
    float weights[16];
    int count;
};

__global__ __launch_bounds__(256) void temporal_average_kernel(TemporalArgs a, long n, uint8_t* out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float r = 0.f;
        for (int k = 0; k < a.count; ++k) r = __fadd_rn(r, __fmul_rn((float)a.frames[k][i], a.weights[k]));
        out[i] = (uint8_t)(int)r;
    }
}

void launch_temporal_average(const uint8_t* const* frames, const float* weights, int count, long n, uint8_t* out,
                             hipStream_t st) {
    if (count < 1 || count > 16) throw Error(1, "temporal_average: window must be 1..16 frames");
    TemporalArgs a{};
    a.count = count;
    for (int k = 0; k < count; ++k) {
        a.frames[k] = frames[k];
        a.weights[k] = weights[k];
    }
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(temporal_average_kernel, dim3(blocks), dim3(256), 0, st, a, n, out);
    FW_HIP_CHECK(hipGetLastError());
}

// tap_denoise.py:614-618: blended = original * (1 - s) + denoised * s (float32, separate roundings); astype(uint8)
__global__ __launch_bounds__(256) void strength_blend_kernel(const uint8_t* __restrict__ orig, const uint8_t* __restrict__ den,
                                                             float one_minus_s, float s, long n, uint8_t* out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (uint8_t)(int)__fadd_rn(__fmul_rn((float)orig[i], one_minus_s), __fmul_rn((float)den[i], s));
}

void launch_strength_blend(const uint8_t* orig, const uint8_t* den, float one_minus_s, float s, long n, uint8_t* out,
                           hipStream_t st) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(strength_blend_kernel, dim3(blocks), dim3(256), 0, st, orig, den, one_minus_s, s, n, out);
    FW_HIP_CHECK(hipGetLastError());
}

// ---- cv2.resize(..., interpolation=cv2.INTER_LANCZOS4) on 8-bit images ---------------------------------------------
// RealESRGANer.enhance's last step when outscale != netscale (pip realesrgan, call site reference
// src/framewright/processors/pytorch_realesrgan.py:223 `upsampler.enhance(img, outscale=config.scale_factor)`): the uint8
// output is resized to (int(w*outscale), int(h*outscale)).  OpenCV's 8-bit path is fixed point: 8 horizontal and 8 vertical
// taps, coefficients = saturate_cast<short>(c * 2048) of the normalised float Lanczos-4 weights, source indices clamped to
// the image, result = saturate_cast<uchar>((sum + 2^21) >> 22).  The horizontal sums stay exact ints, so one fused pass over
// the 8x8 footprint equals OpenCV's two passes bit for bit.  Tables come from the host (double sin/cos as OpenCV computes
// them; a device libm could differ in the last place).
__global__ __launch_bounds__(256) void resize_lanczos4_u8_kernel(const uint8_t* __restrict__ src, int Hs, int Ws, int C,
                                                                 uint8_t* __restrict__ dst, int Hd, int Wd,
                                                                 const int* __restrict__ xofs, const short* __restrict__ ialpha,
                                                                 const int* __restrict__ yofs, const short* __restrict__ ibeta) {
    const long total = (long)Hd * Wd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int dy = (int)(i / Wd), dx = (int)(i - (long)dy * Wd);
        const int sx = xofs[dx], sy = yofs[dy];
        int xs[8];
        short a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int x = sx + j - 3;
            x = x < 0 ? 0 : (x >= Ws ? Ws - 1 : x);
            xs[j] = x * C;
            a[j] = ialpha[dx * 8 + j];
            b[j] = ibeta[dy * 8 + j];
        }
        for (int c = 0; c < C; ++c) {
            int v = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int y = sy + k - 3;
                y = y < 0 ? 0 : (y >= Hs ? Hs - 1 : y);
                const uint8_t* row = src + (long)y * Ws * C + c;
                int h = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) h += (int)row[xs[j]] * a[j];
                v += h * b[k];
            }
            v = (v + (1 << 21)) >> 22;
            dst[i * C + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

// OpenCV interpolateLanczos4 (imgproc, resize.cpp): weights of the 8 taps at fractional position x
static void lanczos4_coeffs(float x, float* coeffs) {
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    if (x < 1.1920928955078125e-7f) {  // FLT_EPSILON
        for (int i = 0; i < 8; i++) coeffs[i] = 0;
        coeffs[3] = 1;
        return;
    }
    float sum = 0;
    const double y0 = -(x + 3) * 3.1415926535897932384626433832795 * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
    for (int i = 0; i < 8; i++) {
        const double y = -(x + 3 - i) * 3.1415926535897932384626433832795 * 0.25;
        coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += coeffs[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) coeffs[i] *= sum;
}

static void lanczos4_tables(int ssize, int dsize, std::vector<int>& ofs, std::vector<short>& coef) {
    const double inv_scale = (double)dsize / ssize, scale = 1. / inv_scale;
    ofs.resize(dsize);
    coef.resize((size_t)dsize * 8);
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        const int s0 = (int)std::floor(f);
        f -= s0;
        ofs[d] = s0;
        float cbuf[8];
        lanczos4_coeffs(f, cbuf);
        for (int k = 0; k < 8; ++k) {
            // saturate_cast<short>(float): round to nearest even, clamp
            const float v = cbuf[k] * 2048.f;
            long r = std::lrintf(v);
            coef[(size_t)d * 8 + k] = (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
        }
    }
}

void launch_resize_lanczos4_u8(const uint8_t* src, int Hs, int Ws, int C, uint8_t* dst, int Hd, int Wd, hipStream_t st) {
    std::vector<int> xofs, yofs;
    std::vector<short> ia, ib;
    lanczos4_tables(Ws, Wd, xofs, ia);
    lanczos4_tables(Hs, Hd, yofs, ib);
    const size_t nx = (size_t)Wd, ny = (size_t)Hd;
    // one device block: [xofs | yofs | ialpha | ibeta]
    const size_t bytes = (nx + ny) * 4 + (nx + ny) * 16;
    char* d = nullptr;
    FW_HIP_CHECK(hipMalloc((void**)&d, bytes));
    std::vector<char> h(bytes);
    memcpy(h.data(), xofs.data(), nx * 4);
    memcpy(h.data() + nx * 4, yofs.data(), ny * 4);
    memcpy(h.data() + (nx + ny) * 4, ia.data(), nx * 16);
    memcpy(h.data() + (nx + ny) * 4 + nx * 16, ib.data(), ny * 16);
    hipError_t e = hipMemcpyAsync(d, h.data(), bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const long total = (long)Hd * Wd;
        const int blocks = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
        hipLaunchKernelGGL(resize_lanczos4_u8_kernel, dim3(blocks), dim3(256), 0, st, src, Hs, Ws, C, dst, Hd, Wd,
                           reinterpret_cast<const int*>(d), reinterpret_cast<const short*>(d + (nx + ny) * 4),
                           reinterpret_cast<const int*>(d + nx * 4), reinterpret_cast<const short*>(d + (nx + ny) * 4 + nx * 16));
        e = hipGetLastError();
    }
    // the tables live in pageable host memory and a per-call device block: finish before both go away (this is the
    // once-per-frame tail of a non-default configuration, not the hot path)
    const hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d);
    FW_HIP_CHECK(e);
    FW_HIP_CHECK(e2);
}

// The same resize on 16-bit images (RealESRGANer.enhance with outscale != netscale on a 16-bit frame).  OpenCV's path for ushort is
// float: HResizeLanczos4<ushort, float, float> then VResizeLanczos4<ushort, float, float, Cast<float, ushort>> - the normalised
// float weights themselves, eight products summed left to right per pass, saturate_cast<ushort>(cvRound(sum)).  This file is
// compiled with -ffp-contract=off, so a * b + c rounds twice like the scalar C++ does.  (OpenCV's SIMD build may associate the
// vertical sum differently; without cv2 in the image the restatement is unpinned either way - oracle/lanczos_ref.py.)
__global__ __launch_bounds__(256) void resize_lanczos4_u16_kernel(const uint16_t* __restrict__ src, int Hs, int Ws, int C,
                                                                  uint16_t* __restrict__ dst, int Hd, int Wd,
                                                                  const int* __restrict__ xofs, const float* __restrict__ alpha,
                                                                  const int* __restrict__ yofs, const float* __restrict__ beta) {
    const long total = (long)Hd * Wd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int dy = (int)(i / Wd), dx = (int)(i - (long)dy * Wd);
        const int sx = xofs[dx], sy = yofs[dy];
        int xs[8];
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int x = sx + j - 3;
            x = x < 0 ? 0 : (x >= Ws ? Ws - 1 : x);
            xs[j] = x * C;
            a[j] = alpha[dx * 8 + j];
            b[j] = beta[dy * 8 + j];
        }
        for (int c = 0; c < C; ++c) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int y = sy + k - 3;
                y = y < 0 ? 0 : (y >= Hs ? Hs - 1 : y);
                const uint16_t* row = src + (long)y * Ws * C + c;
                float h = (float)row[xs[0]] * a[0];
#pragma unroll
                for (int j = 1; j < 8; ++j) h = h + (float)row[xs[j]] * a[j];
                v = k == 0 ? h * b[0] : v + h * b[k];
            }
            const float r = rintf(v);   // cvRound: round half to even
            dst[i * C + c] = (uint16_t)(r < 0.f ? 0.f : (r > 65535.f ? 65535.f : r));
        }
    }
}

void launch_resize_lanczos4_u16(const uint16_t* src, int Hs, int Ws, int C, uint16_t* dst, int Hd, int Wd, hipStream_t st) {
    auto tables = [](int ssize, int dsize, std::vector<int>& ofs, std::vector<float>& coef) {
        const double inv_scale = (double)dsize / ssize, scale = 1. / inv_scale;
        ofs.resize(dsize);
        coef.resize((size_t)dsize * 8);
        for (int d = 0; d < dsize; ++d) {
            float f = (float)((d + 0.5) * scale - 0.5);
            const int s0 = (int)std::floor(f);
            f -= s0;
            ofs[d] = s0;
            lanczos4_coeffs(f, coef.data() + (size_t)d * 8);
        }
    };
    std::vector<int> xofs, yofs;
    std::vector<float> fa, fb;
    tables(Ws, Wd, xofs, fa);
    tables(Hs, Hd, yofs, fb);
    const size_t nx = (size_t)Wd, ny = (size_t)Hd;
    const size_t bytes = (nx + ny) * 4 + (nx + ny) * 32;   // [xofs | yofs | alpha | beta]
    char* d = nullptr;
    FW_HIP_CHECK(hipMalloc((void**)&d, bytes));
    std::vector<char> h(bytes);
    memcpy(h.data(), xofs.data(), nx * 4);
    memcpy(h.data() + nx * 4, yofs.data(), ny * 4);
    memcpy(h.data() + (nx + ny) * 4, fa.data(), nx * 32);
    memcpy(h.data() + (nx + ny) * 4 + nx * 32, fb.data(), ny * 32);
    hipError_t e = hipMemcpyAsync(d, h.data(), bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const long total = (long)Hd * Wd;
        const int blocks = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
        hipLaunchKernelGGL(resize_lanczos4_u16_kernel, dim3(blocks), dim3(256), 0, st, src, Hs, Ws, C, dst, Hd, Wd,
                           reinterpret_cast<const int*>(d), reinterpret_cast<const float*>(d + (nx + ny) * 4),
                           reinterpret_cast<const int*>(d + nx * 4), reinterpret_cast<const float*>(d + (nx + ny) * 4 + nx * 32));
        e = hipGetLastError();
    }
    const hipError_t e2 = hipStreamSynchronize(st);   // tables in pageable host memory and a per-call device block (as for 8-bit)
    (void)hipFree(d);
    FW_HIP_CHECK(e);
    FW_HIP_CHECK(e2);
}

// ---- cv2.resize(img, (w, h)) with the default interpolation (INTER_LINEAR) on 8-bit images ------------------------------------
// `_paste_face_back` of the reference's AESRGANFaceRestorer (src/framewright/processors/aesrgan_face.py:553) brings the enhanced crop
// back to the size of its region with it.  OpenCV's 8-bit path (imgproc/resize.cpp; restated in oracle/face_ref.py, parity unpinned:
// no cv2 in the image): an exact 2 x 2 decimation - the default upscale_factor 2 - takes the "area fast" path, (a + b + c + d + 2) >> 2;
// everything else is the fixed-point bilinear: 11-bit coefficients from float fx = (dx + 0.5) * scale - 0.5, both borders clamped
// with fx = 0, the horizontal sums kept as ints, the vertical pass (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
__global__ __launch_bounds__(256) void resize_area2_u8_kernel(const uint8_t* __restrict__ src, int Ws, int C, uint8_t* __restrict__ dst, int Hd,
                                                              int Wd) {
    const long total = (long)Hd * Wd * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long px = i / C;
        const int dy = (int)(px / Wd), dx = (int)(px - (long)dy * Wd);
        const uint8_t* s0 = src + ((long)(2 * dy) * Ws + 2 * dx) * C + c;
        const uint8_t* s1 = s0 + (long)Ws * C;
        dst[i] = (uint8_t)(((int)s0[0] + (int)s0[C] + (int)s1[0] + (int)s1[C] + 2) >> 2);
    }
}

__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* __restrict__ src, int Hs, int Ws, int C, uint8_t* __restrict__ dst,
                                                               int Hd, int Wd, const int* __restrict__ xofs, const short* __restrict__ ialpha,
                                                               const int* __restrict__ yofs, const short* __restrict__ ibeta) {
    const long total = (long)Hd * Wd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int dy = (int)(i / Wd), dx = (int)(i - (long)dy * Wd);
        const int sx = xofs[dx], sy = yofs[dy];
        const int sx1 = sx + 1 < Ws ? sx + 1 : Ws - 1, sy1 = sy + 1 < Hs ? sy + 1 : Hs - 1;
        const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1], b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        const uint8_t* r0 = src + (long)sy * Ws * C;
        const uint8_t* r1 = src + (long)sy1 * Ws * C;
        for (int c = 0; c < C; ++c) {
            const int h0 = (int)r0[sx * C + c] * a0 + (int)r0[sx1 * C + c] * a1;
            const int h1 = (int)r1[sx * C + c] * a0 + (int)r1[sx1 * C + c] * a1;
            const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            dst[i * C + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

static void linear_tables(int ssize, int dsize, std::vector<int>& ofs, std::vector<short>& coef) {
    const double inv_scale = (double)dsize / ssize, scale = 1. / inv_scale;
    ofs.resize(dsize);
    coef.resize((size_t)dsize * 2);
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s0 = (int)std::floor(f);
        f -= s0;
        if (s0 < 0) f = 0.f, s0 = 0;
        if (s0 >= ssize - 1) f = 0.f, s0 = ssize - 1;
        ofs[d] = s0;
        coef[(size_t)d * 2] = (short)std::lrintf((1.f - f) * 2048.f);   // saturate_cast<short>: round to nearest even; |c| <= 2048
        coef[(size_t)d * 2 + 1] = (short)std::lrintf(f * 2048.f);
    }
}

void launch_resize_linear_u8(const uint8_t* src, int Hs, int Ws, int C, uint8_t* dst, int Hd, int Wd, hipStream_t st) {
    const long total = (long)Hd * Wd;
    const int blocks = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    if (Hs == Hd && Ws == Wd) {
        FW_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)total * C, hipMemcpyDeviceToDevice, st));
        return;
    }
    if (Ws == 2 * Wd && Hs == 2 * Hd) {
        hipLaunchKernelGGL(resize_area2_u8_kernel, dim3(blocks), dim3(256), 0, st, src, Ws, C, dst, Hd, Wd);
        FW_HIP_CHECK(hipGetLastError());
        return;
    }
    std::vector<int> xofs, yofs;
    std::vector<short> ia, ib;
    linear_tables(Ws, Wd, xofs, ia);
    linear_tables(Hs, Hd, yofs, ib);
    const size_t nx = (size_t)Wd, ny = (size_t)Hd;
    const size_t bytes = (nx + ny) * 4 + (nx + ny) * 4;   // [xofs | yofs | ialpha | ibeta]
    char* d = nullptr;
    FW_HIP_CHECK(hipMalloc((void**)&d, bytes));
    std::vector<char> h(bytes);
    memcpy(h.data(), xofs.data(), nx * 4);
    memcpy(h.data() + nx * 4, yofs.data(), ny * 4);
    memcpy(h.data() + (nx + ny) * 4, ia.data(), nx * 4);
    memcpy(h.data() + (nx + ny) * 4 + nx * 4, ib.data(), ny * 4);
    hipError_t e = hipMemcpyAsync(d, h.data(), bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(resize_linear_u8_kernel, dim3(blocks), dim3(256), 0, st, src, Hs, Ws, C, dst, Hd, Wd, reinterpret_cast<const int*>(d),
                           reinterpret_cast<const short*>(d + (nx + ny) * 4), reinterpret_cast<const int*>(d + nx * 4),
                           reinterpret_cast<const short*>(d + (nx + ny) * 4 + nx * 4));
        e = hipGetLastError();
    }
    // the tables live in pageable host memory and a per-call device block: finish before both go away (a face region per call)
    const hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d);
    FW_HIP_CHECK(e);
    FW_HIP_CHECK(e2);
}

// ---- the blend of `_paste_face_back` (aesrgan_face.py:556-584), float32 statement for statement -----------------------------------
// mask = 1, then for i < feather = min(w, h) // 8: rows i and h-1-i and columns i and w-1-i are multiplied by float32(i / feather)
// (i / feather in double); region = orig * (1 - mask * s) + enh * mask * s in float32 (no FMA: this file is built with
// -ffp-contract=off), truncating cast.  `frame` is updated in place inside [y1, y2) x [x1, x2).
__global__ __launch_bounds__(256) void face_paste_u8_kernel(uint8_t* frame, int W, int x1, int y1, int rw, int rh, const uint8_t* __restrict__ enh,
                                                            float s, int feather) {
    const long total = (long)rh * rw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / rw), x = (int)(i - (long)y * rw);
        float m = 1.0f;
        if (feather > 0) {
            if (y < feather) m = m * (float)((double)y / (double)feather);
            if (y >= rh - feather) m = m * (float)((double)(rh - 1 - y) / (double)feather);
            if (x < feather) m = m * (float)((double)x / (double)feather);
            if (x >= rw - feather) m = m * (float)((double)(rw - 1 - x) / (double)feather);
        }
        const float t2 = 1.0f - m * s;
        uint8_t* o = frame + ((long)(y1 + y) * W + (x1 + x)) * 3;
        const uint8_t* e = enh + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = (float)o[c] * t2;
            const float b = ((float)e[c] * m) * s;
            o[c] = (uint8_t)(int)(a + b);
        }
    }
}

void launch_face_paste_u8(uint8_t* frame, int H, int W, int x1, int y1, int x2, int y2, const uint8_t* enh, float strength, hipStream_t st) {
    const int rw = x2 - x1, rh = y2 - y1;
    if (rw <= 0 || rh <= 0 || x1 < 0 || y1 < 0 || x2 > W || y2 > H) throw Error(1, "face_paste: region outside the frame");
    const int feather = (rw < rh ? rw : rh) / 8;
    const long total = (long)rw * rh;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(face_paste_u8_kernel, dim3(blocks), dim3(256), 0, st, frame, W, x1, y1, rw, rh, enh, strength, feather);
    FW_HIP_CHECK(hipGetLastError());
}

// ---- grain add-back (reference src/framewright/processors/tap_denoise.py:621-632, :1015-1023) -------------------------
//   gray = cv2.cvtColor(orig, COLOR_BGR2GRAY); blurred = cv2.GaussianBlur(gray, (0, 0), 3); grain = cv2.subtract(gray, blurred)
//   denoised = cv2.add(denoised, (GRAY2BGR(grain) * factor).astype(np.uint8))
// restated from OpenCV's published 8-bit arithmetic (cv2 is absent here: unpinned): BGR2GRAY = (1868 B + 9617 G + 4899 R +
// 2^13) >> 14; GaussianBlur of an 8-bit image with sigma 3 takes ksize 19 and the bit-exact fixed-point path - the float
// kernel converted to 8 fraction bits with error diffusion so that it sums to 256 (grain_kernel19 below), horizontal sums
// exact in 8.8, vertical sums exact in 16.16, result (sum + 2^15) >> 16, BORDER_REFLECT_101; subtract / add saturate; the
// scaling is a float64 product truncated to uint8.
struct GrainKernel {
    int k[19];
};

__device__ __forceinline__ int gray_of(const uint8_t* px) { return (px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + (1 << 13)) >> 14; }

__global__ __launch_bounds__(256) void grain_hblur_kernel(const uint8_t* __restrict__ bgr, int H, int W, uint16_t* hbuf, GrainKernel gk) {
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 19; ++j) acc += gk.k[j] * gray_of(bgr + ((long)y * W + reflect101(x + j - 9, W)) * 3);
        hbuf[i] = (uint16_t)acc;   // <= 255 * 256
    }
}

__global__ __launch_bounds__(256) void grain_add_kernel(const uint8_t* __restrict__ orig, const uint16_t* __restrict__ hbuf,
                                                        const uint8_t* __restrict__ den, int H, int W, double factor,
                                                        uint8_t* out, GrainKernel gk) {
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 19; ++j) acc += gk.k[j] * (int)hbuf[(long)reflect101(y + j - 9, H) * W + x];
        int blurred = (acc + (1 << 15)) >> 16;
        blurred = blurred > 255 ? 255 : blurred;
        int grain = gray_of(orig + i * 3) - blurred;
        grain = grain < 0 ? 0 : grain;
        const int add = (int)(uint8_t)(long)((double)grain * factor);   // float64 product, astype(uint8)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int v = den[i * 3 + c] + add;
            out[i * 3 + c] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
}

// getGaussianKernelBitExact(19, 3) + getGaussianKernelFixedPoint_ED(8 fraction bits)
static GrainKernel grain_kernel19() {
    constexpr int n = 19;
    const double sigma = 3.0, scale2x = -0.125 / (sigma * sigma);
    double v[n / 2], sum = 0;
    for (int i = 0, x = 1 - n; i < n / 2; ++i, x += 2) {
        v[i] = std::exp((double)(x * x) * scale2x);
        sum += v[i];
    }
    sum = sum * 2 + 1;
    const double mul = 1.0 / sum;
    GrainKernel g;
    double err = 0;
    long tot = 0;
    for (int i = 0; i < n / 2; ++i) {
        const double adj = v[i] * mul * 256.0 + err;
        const long v0 = std::lrint(adj);   // cvRound: to nearest even
        err = adj - (double)v0;
        g.k[i] = g.k[n - 1 - i] = (int)v0;
        tot += v0;
    }
    g.k[n / 2] = (int)(256 - 2 * tot);
    return g;
}

void launch_grain_addback(const uint8_t* orig, const uint8_t* den, int H, int W, double factor, uint16_t* tmp, uint8_t* out,
                          hipStream_t st) {
    static const GrainKernel gk = grain_kernel19();
    const long n = (long)H * W;
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(grain_hblur_kernel, dim3(blocks), dim3(256), 0, st, orig, H, W, tmp, gk);
    hipLaunchKernelGGL(grain_add_kernel, dim3(blocks), dim3(256), 0, st, orig, tmp, den, H, W, factor, out, gk);
    FW_HIP_CHECK(hipGetLastError());
}

// ---- `_preserve_edges` of the classical temporal denoiser (temporal_denoise.py:1636-1667) ----------------------------------------
//   gray = BGR2GRAY(original); edges = Canny(gray, t, 3t); edges = dilate(edges, 3x3); mask = GaussianBlur(edges / 255, (5, 5), 0)
//   out  = (original * mask + denoised * (1 - mask)).astype(uint8)
// OpenCV's 8-bit algorithms restated (oracle/temporal_ref.py has the same statement in numpy and the citations): 14-bit gray
// weights, Sobel 3x3 with replicated borders, L1 magnitude, the fixed-point tangent test of the non-maximum suppression,
// hysteresis, 3x3 maximum, the fixed [1 4 6 4 1] / 16 float32 kernel with BORDER_REFLECT_101.  This file is compiled without FMA
// contraction, so the float32 passes round exactly like numpy's.
__device__ __forceinline__ int pe_gray(const uint8_t* __restrict__ bgr, int H, int W, int y, int x) {
    y = min(max(y, 0), H - 1);   // BORDER_REPLICATE of the Sobel
    x = min(max(x, 0), W - 1);
    const uint8_t* p = bgr + ((long)y * W + x) * 3;
    return (p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14;
}

// dx, dy (int16) and the L1 magnitude of every pixel
__global__ __launch_bounds__(256) void pe_sobel_kernel(const uint8_t* __restrict__ bgr, int H, int W, short* __restrict__ dxy,
                                                       short* __restrict__ mag) {
    const long n = (long)H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        int g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = pe_gray(bgr, H, W, y + a - 1, x + b - 1);
        const int dx = (g[0][2] + 2 * g[1][2] + g[2][2]) - (g[0][0] + 2 * g[1][0] + g[2][0]);
        const int dy = (g[2][0] + 2 * g[2][1] + g[2][2]) - (g[0][0] + 2 * g[0][1] + g[0][2]);
        dxy[2 * i] = (short)dx;
        dxy[2 * i + 1] = (short)dy;
        mag[i] = (short)(abs(dx) + abs(dy));
    }
}

// non-maximum suppression: map = 2 (above `high`: an edge), 0 (above `low`: an edge if connected to one), 1 (no edge)
__global__ __launch_bounds__(256) void pe_nms_kernel(const short* __restrict__ dxy, const short* __restrict__ mag, int H, int W, int lo,
                                                     int hi, uint8_t* __restrict__ map) {
    const long n = (long)H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        auto M = [&](int yy, int xx) { return (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0 : (int)mag[(long)yy * W + xx]; };
        const int m = mag[i], xs = dxy[2 * i], ys = dxy[2 * i + 1];
        uint8_t v = 1;
        if (m > lo) {
            const long ax = abs(xs), ay = (long)abs(ys) << 15;
            const long tg22 = ax * 13573;
            bool keep;
            if (ay < tg22) {
                keep = m > M(y, x - 1) && m >= M(y, x + 1);
            } else if (ay > tg22 + (ax << 16)) {
                keep = m > M(y - 1, x) && m >= M(y + 1, x);
            } else {
                const int s = (xs ^ ys) < 0 ? -1 : 1;
                keep = m > M(y - 1, x - s) && m > M(y + 1, x + s);
            }
            if (keep) v = m > hi ? 2 : 0;
        }
        map[i] = v;
    }
}

// hysteresis: one 32 x 32 tile (with a one-pixel ring) per workgroup is grown to its fixed point in LDS; the host repeats the
// launch until no tile changed.  Only 0 -> 2 transitions exist, so the fixed point does not depend on the order of the sweeps.
__global__ __launch_bounds__(256) void pe_hysteresis_kernel(uint8_t* __restrict__ map, int H, int W, int* __restrict__ changed) {
    __shared__ uint8_t t[34][36];
    const int tiles_x = (W + 31) / 32;
    const int ty0 = (blockIdx.x / tiles_x) * 32, tx0 = (blockIdx.x % tiles_x) * 32;
    for (int k = threadIdx.x; k < 34 * 34; k += 256) {
        const int yy = k / 34, xx = k - yy * 34, gy = ty0 + yy - 1, gx = tx0 + xx - 1;
        t[yy][xx] = (gy < 0 || gy >= H || gx < 0 || gx >= W) ? 1 : map[(long)gy * W + gx];
    }
    __syncthreads();
    bool any = false;
    for (int it = 0; it < 1024; ++it) {
        bool ch = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = threadIdx.x + 256 * q, yy = 1 + k / 32, xx = 1 + (k & 31);
            if (t[yy][xx] == 0) {
                bool nb = false;
#pragma unroll
                for (int a = -1; a <= 1; ++a)
#pragma unroll
                    for (int b = -1; b <= 1; ++b) nb |= t[yy + a][xx + b] == 2;
                if (nb) {
                    t[yy][xx] = 2;
                    ch = true;
                }
            }
        }
        if (!__syncthreads_or(ch)) break;
        any = true;
    }
    if (any) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = threadIdx.x + 256 * q, yy = 1 + k / 32, xx = 1 + (k & 31), gy = ty0 + yy - 1, gx = tx0 + xx - 1;
            if (gy < H && gx < W) map[(long)gy * W + gx] = t[yy][xx];
        }
        if (threadIdx.x == 0) atomicOr(changed, 1);
    }
}

// 3x3 maximum of the edge image (map == 2), then the horizontal pass of the 5-tap Gaussian on edges / 255 (0.0f or 1.0f)
__global__ __launch_bounds__(256) void pe_mask_rows_kernel(const uint8_t* __restrict__ map, int H, int W, float* __restrict__ rows) {
    const long n = (long)H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        auto refl = [](int v, int len) { return len == 1 ? 0 : (v < 0 ? -v : (v >= len ? 2 * len - 2 - v : v)); };
        auto dil = [&](int xx) {   // dilated edge value at (y, xx), xx inside the image
            for (int a = -1; a <= 1; ++a)
                for (int b = -1; b <= 1; ++b) {
                    const int yy = y + a, x2 = xx + b;
                    if (yy >= 0 && yy < H && x2 >= 0 && x2 < W && map[(long)yy * W + x2] == 2) return 1.0f;
                }
            return 0.0f;
        };
        float c = dil(x);
        float l1 = dil(min(max(refl(x - 1, W), 0), W - 1)), r1 = dil(min(max(refl(x + 1, W), 0), W - 1));
        float l2 = dil(min(max(refl(x - 2, W), 0), W - 1)), r2 = dil(min(max(refl(x + 2, W), 0), W - 1));
        rows[i] = (c * 0.375f + (l1 + r1) * 0.25f) + ((l2 + r2) * 0.0625f);
    }
}

// vertical pass + blend
__global__ __launch_bounds__(256) void pe_blend_kernel(const uint8_t* __restrict__ orig, const uint8_t* __restrict__ den,
                                                       const float* __restrict__ rows, int H, int W, uint8_t* __restrict__ out) {
    const long n = (long)H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        auto R = [&](int yy) {
            yy = H == 1 ? 0 : (yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy));
            yy = min(max(yy, 0), H - 1);
            return rows[(long)yy * W + x];
        };
        const float mask = (R(y) * 0.375f + (R(y - 1) + R(y + 1)) * 0.25f) + ((R(y - 2) + R(y + 2)) * 0.0625f);
        const float inv = 1.0f - mask;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (float)orig[i * 3 + c] * mask + (float)den[i * 3 + c] * inv;
            out[i * 3 + c] = (uint8_t)v;
        }
    }
}

size_t preserve_edges_scratch_bytes(int H, int W) {
    const size_t n = (size_t)H * W;
    return n * (4 + 2 + 1 + 4) + 256 + 64 + 64;   // two alignment gaps (<= 255, <= 63 bytes) and the round flags
}

void launch_preserve_edges(const uint8_t* orig, const uint8_t* den, int H, int W, int lo, int hi, void* scratch, uint8_t* out,
                           hipStream_t st) {
    const size_t n = (size_t)H * W;
    char* s = (char*)scratch;
    short* dxy = (short*)s;
    short* mag = (short*)(s + n * 4);
    float* rows = (float*)(s + ((n * 6 + 255) / 256) * 256);
    uint8_t* map = (uint8_t*)(rows + n);
    int* changed = (int*)(((size_t)(map + n) + 63) / 64 * 64);
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(pe_sobel_kernel, dim3(blocks), dim3(256), 0, st, orig, H, W, dxy, mag);
    hipLaunchKernelGGL(pe_nms_kernel, dim3(blocks), dim3(256), 0, st, dxy, mag, H, W, lo, hi, map);
    const int tiles = ((W + 31) / 32) * ((H + 31) / 32);
    // Global sweeps until no tile changed.  A chain of weak pixels that crosses T tiles needs T sweeps; almost every frame is done
    // after one or two, so the sweeps go out four at a time behind ONE host round trip (a sweep over a converged map is a no-op
    // that reports no change).  A map that has not converged after CAP sweeps is an error, not a silently wrong mask.
    constexpr int BATCH = 4, CAP = 4096;
    bool converged = false;
    for (int round = 0; round < CAP && !converged; round += BATCH) {
        FW_HIP_CHECK(hipMemsetAsync(changed, 0, BATCH * sizeof(int), st));
        for (int b = 0; b < BATCH; ++b) hipLaunchKernelGGL(pe_hysteresis_kernel, dim3(tiles), dim3(256), 0, st, map, H, W, changed + b);
        int flags[BATCH] = {};
        FW_HIP_CHECK(hipMemcpyAsync(flags, changed, BATCH * sizeof(int), hipMemcpyDeviceToHost, st));
        FW_HIP_CHECK(hipStreamSynchronize(st));
        converged = flags[BATCH - 1] == 0;
    }
    if (!converged) throw Error(4, "preserve_edges: the hysteresis did not reach its fixed point");
    hipLaunchKernelGGL(pe_mask_rows_kernel, dim3(blocks), dim3(256), 0, st, map, H, W, rows);
    hipLaunchKernelGGL(pe_blend_kernel, dim3(blocks), dim3(256), 0, st, orig, den, rows, H, W, out);
    FW_HIP_CHECK(hipGetLastError());
}

}  // namespace fw
