// IFNet (RIFE v4.6) building blocks that are not convolutions (kernel set K7 of SURVEY.md §8a): bilinear resize,
// backward warp (grid_sample bilinear / border / align_corners=True), sigmoid-mask blend, pixel (un)shuffle glue.
//
// The reference only shells out to the external binary `rife-ncnn-vulkan` (reference
// src/framewright/processors/interpolation.py:628-650); the arithmetic restated here is IFNet_HDv3 v4.6 as recorded in
// SURVEY.md §A.5 and in oracle/ifnet_ref.py ("parity vs upstream unpinned").  All tensors are fp32 NHWC with a handful
// of channels (image 3, flow 4, mask 1): HBM-bound byte work, one thread per output pixel, no LDS.
#include "fw_internal.h"
#include "../../include/framewright_hip.h"

namespace fw {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T>
__device__ __forceinline__ T cvt(float f);
template <>
__device__ __forceinline__ __bf16 cvt<__bf16>(float f) { return (__bf16)f; }
template <>
__device__ __forceinline__ _Float16 cvt<_Float16>(float f) { return (_Float16)f; }

// uint8 BGR H x W -> fp32 RGB/255 [Hp][Wp][3], zero outside (Practical-RIFE pads bottom/right to a multiple of 32)
__global__ __launch_bounds__(256) void u8_to_rgb_f32_kernel(const uint8_t* __restrict__ in, int H, int W, int Hp, int Wp,
                                                                   float* out) {
    const long n = (long)Hp * Wp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / Wp), x = (int)(i - (long)y * Wp);
        float r = 0, g = 0, b = 0;
        if (y < H && x < W) {
            const uint8_t* px = in + ((size_t)y * W + x) * 3;
            b = px[0] / 255.0f;
            g = px[1] / 255.0f;
            r = px[2] / 255.0f;
        }
        out[i * 3 + 0] = r;
        out[i * 3 + 1] = g;
        out[i * 3 + 2] = b;
    }
}

// torch F.interpolate(mode="bilinear", align_corners=False, scale_factor=sf): src = (dst + 0.5) / sf - 0.5, clamped at 0
__device__ __forceinline__ void bilin_setup(int d, float inv_sf, int n, int* i0, int* i1, float* w1) {
    float s = ((float)d + 0.5f) * inv_sf - 0.5f;
    if (s < 0.f) s = 0.f;
    int a = (int)s;
    if (a > n - 1) a = n - 1;
    *i0 = a;
    *i1 = a + 1 < n ? a + 1 : n - 1;
    *w1 = s - (float)a;
}

// dst[y][x][dst_coff + c] = mul * bilinear(src[..][c]), c < C
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ src, int Hs, int Ws, int C, float* dst,
                                                              int Hd, int Wd, int dst_cstride, int dst_coff, float inv_sf,
                                                              float mul) {
    const long n = (long)Hd * Wd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / Wd), x = (int)(i - (long)y * Wd);
        int y0, y1, x0, x1;
        float wy, wx;
        bilin_setup(y, inv_sf, Hs, &y0, &y1, &wy);
        bilin_setup(x, inv_sf, Ws, &x0, &x1, &wx);
        const float* p00 = src + ((size_t)y0 * Ws + x0) * C;
        const float* p01 = src + ((size_t)y0 * Ws + x1) * C;
        const float* p10 = src + ((size_t)y1 * Ws + x0) * C;
        const float* p11 = src + ((size_t)y1 * Ws + x1) * C;
        float* o = dst + (size_t)i * dst_cstride + dst_coff;
        for (int c = 0; c < C; ++c) {
            const float top = p00[c] * (1.f - wx) + p01[c] * wx;
            const float bot = p10[c] * (1.f - wx) + p11[c] * wx;
            o[c] = (top * (1.f - wy) + bot * wy) * mul;
        }
    }
}

// grid_sample(img, base + flow, bilinear, padding_mode='border', align_corners=True) with flow in pixels:
// sample position (x + fx, y + fy) clamped to the image, then bilinear.
__device__ __forceinline__ void warp_px(const float* __restrict__ img, int H, int W, float sx, float sy, float* o) {
    sx = fminf(fmaxf(sx, 0.f), (float)(W - 1));
    sy = fminf(fmaxf(sy, 0.f), (float)(H - 1));
    const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
    const int x1 = x0 + 1 < W ? x0 + 1 : W - 1, y1 = y0 + 1 < H ? y0 + 1 : H - 1;
    const float wx = sx - (float)x0, wy = sy - (float)y0;
    const float* p00 = img + ((size_t)y0 * W + x0) * 3;
    const float* p01 = img + ((size_t)y0 * W + x1) * 3;
    const float* p10 = img + ((size_t)y1 * W + x0) * 3;
    const float* p11 = img + ((size_t)y1 * W + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        o[c] = (p00[c] * (1.f - wx) + p01[c] * wx) * (1.f - wy) + (p10[c] * (1.f - wx) + p11[c] * wx) * wy;
}

// X = cat(warp(I0, flow[:2]), warp(I1, flow[2:4]), timestep, mask)  (8 ch), or cat(I0, I1, timestep) (7 ch) when flow == 0
__global__ __launch_bounds__(256) void ifnet_build_x_kernel(const float* __restrict__ i0, const float* __restrict__ i1,
                                                            const float* __restrict__ flow, const float* __restrict__ mask,
                                                            int H, int W, float timestep, float* X) {
    const long n = (long)H * W;
    const int C = flow ? 8 : 7;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        float* o = X + (size_t)i * C;
        if (flow) {
            const float* f = flow + (size_t)i * 4;
            warp_px(i0, H, W, x + f[0], y + f[1], o);
            warp_px(i1, H, W, x + f[2], y + f[3], o + 3);
            o[6] = timestep;
            o[7] = mask[i];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                o[c] = i0[i * 3 + c];
                o[3 + c] = i1[i * 3 + c];
            }
            o[6] = timestep;
        }
    }
}

// pixel_unshuffle(2) + cast: src [h][w][C] (fp32 or operand-typed) -> dst typed [h/2][w/2][Cpad], channel c*4 + dy*2 + dx,
// zero for channels >= 4*C.  (Front end of the stride-2 convs, which run as 3x3 convs on the unshuffled tensor.)
template <typename T, typename S>
__global__ __launch_bounds__(256) void unshuffle_cast_kernel(const S* __restrict__ src, int h, int w, int C, int src_cstride,
                                                             T* dst, int Cpad) {
    const int ho = h / 2, wo = w / 2;
    const long n = (long)ho * wo * Cpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % Cpad);
        const long pix = i / Cpad;
        const int y = (int)(pix / wo), x = (int)(pix - (long)y * wo);
        float v = 0.f;
        if (ch < 4 * C) {
            const int c = ch >> 2, dy = (ch >> 1) & 1, dx = ch & 1;
            v = (float)src[((size_t)(2 * y + dy) * w + (2 * x + dx)) * src_cstride + c];
        }
        dst[i] = cvt<T>(v);
    }
}

// The same for a typed source with an even channel count (between the two stride-2 convs of an IFBlock): a thread owns 8 output channels
// = 2 input channels x 4 sub-pixels: four 4-byte loads and one 16-byte store where the kernel above did eight 2-byte loads and stores.
template <typename T>
__global__ __launch_bounds__(256) void unshuffle_typed8_kernel(const T* __restrict__ src, int h, int w, int C, int src_cstride, T* dst, int Cpad) {
    const int wo = w / 2, groups = Cpad / 8;
    const long n = (long)(h / 2) * wo * groups;
    typedef T t2 __attribute__((ext_vector_type(2)));
    typedef T t8 __attribute__((ext_vector_type(8)));
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        const long pix = i / groups;
        const int y = (int)(pix / wo), x = (int)(pix - (long)y * wo);
        const int c = 2 * g;                      // output channels 8 g .. 8 g + 7 = input channels c, c + 1, sub-pixels 0 .. 3
        t8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)0.f;
        if (c < C) {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const t2 v = *reinterpret_cast<const t2*>(src + ((size_t)(2 * y + (sub >> 1)) * w + (2 * x + (sub & 1))) * src_cstride + c);
                o[sub] = v[0];
                o[4 + sub] = v[1];
            }
        }
        *reinterpret_cast<t8*>(dst + pix * Cpad + 8 * g) = o;
    }
}

// lastconv glue: src fp32 [h][w][cs], channel n = ((c6*4 + qy*2 + qx)*4 + py*2 + px)  ->  dst [4h][4w][6]:
// ConvTranspose2d(k4,s2,p1) computed as a 3x3 conv with 4 parity groups (py,px), followed by PixelShuffle(2) (qy,qx).
__global__ __launch_bounds__(256) void depth_to_space4_kernel(const float* __restrict__ src, int h, int w, int cs, float* dst) {
    const long n = (long)h * w * 96;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 96);
        const long pix = i / 96;
        const int y = (int)(pix / w), x = (int)(pix - (long)y * w);
        const int px = ch & 1, py = (ch >> 1) & 1, co = ch >> 2;
        const int qx = co & 1, qy = (co >> 1) & 1, c6 = co >> 2;
        dst[((size_t)(4 * y + 2 * py + qy) * (4 * w) + (4 * x + 2 * px + qx)) * 6 + c6] = src[pix * cs + ch];
    }
}

// tmp_s [hs][ws][6] (low resolution) -> full resolution by bilinear x scale; flow (+)= tmp[:4] * scale, mask (+)= tmp[4]
__global__ __launch_bounds__(256) void ifnet_accumulate_kernel(const float* __restrict__ tmp, int hs, int ws, int H, int W,
                                                               float scale, float* flow, float* mask, int first) {
    const long n = (long)H * W;
    const float inv_sf = 1.0f / scale;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        int y0, y1, x0, x1;
        float wy, wx;
        bilin_setup(y, inv_sf, hs, &y0, &y1, &wy);
        bilin_setup(x, inv_sf, ws, &x0, &x1, &wx);
        const float* p00 = tmp + ((size_t)y0 * ws + x0) * 6;
        const float* p01 = tmp + ((size_t)y0 * ws + x1) * 6;
        const float* p10 = tmp + ((size_t)y1 * ws + x0) * 6;
        const float* p11 = tmp + ((size_t)y1 * ws + x1) * 6;
        float v[5];
#pragma unroll
        for (int c = 0; c < 5; ++c)
            v[c] = (p00[c] * (1.f - wx) + p01[c] * wx) * (1.f - wy) + (p10[c] * (1.f - wx) + p11[c] * wx) * wy;
#pragma unroll
        for (int c = 0; c < 4; ++c) flow[i * 4 + c] = (first ? 0.f : flow[i * 4 + c]) + v[c] * scale;
        mask[i] = (first ? 0.f : mask[i]) + v[4];
    }
}

// The front of an IFBlock as ONE kernel: build_x -> F.interpolate(x, 1 / s) -> F.interpolate(flow, 1 / s) / s -> cat -> pixel_unshuffle(2)
// -> operand type.  The four kernels above wrote and re-read the 8-channel X at full resolution for every block (at 1080p: 200 of the
// 1820 us of a forward in the last block alone, half of it an identity resize at s = 1).  Here a thread owns one pixel of the RESIZED map
// (4 threads = one pixel of the unshuffled output), evaluates X and the flow only at the full-resolution taps that pixel samples (at
// s = 8 that is 4 of 64 positions) with the SAME expressions in the same order as those kernels (a tap whose weight is exactly 0 is
// skipped: p * 1 + q * 0 = p), and a workgroup's 64 output pixels leave through an LDS tile as contiguous 16-byte pieces (channel
// c * 4 + dy * 2 + dx, zeros behind 4 cin).
struct IfnX {
    float v[12];
};
template <int CIN>
__device__ __forceinline__ IfnX ifnet_x_at(const float* __restrict__ i0, const float* __restrict__ i1, const float* __restrict__ flow,
                                           const float* __restrict__ mask, int H, int W, float timestep, int yy, int xx) {
    IfnX r;
    const size_t ip = (size_t)yy * W + xx;
    if constexpr (CIN == 12) {
        const float4 f = *reinterpret_cast<const float4*>(flow + ip * 4);
        warp_px(i0, H, W, xx + f.x, yy + f.y, r.v);
        warp_px(i1, H, W, xx + f.z, yy + f.w, r.v + 3);
        r.v[6] = timestep;
        r.v[7] = mask[ip];
        r.v[8] = f.x, r.v[9] = f.y, r.v[10] = f.z, r.v[11] = f.w;
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r.v[c] = i0[ip * 3 + c];
            r.v[3 + c] = i1[ip * 3 + c];
        }
        r.v[6] = timestep;
#pragma unroll
        for (int c = 7; c < 12; ++c) r.v[c] = 0.f;
    }
    return r;
}

template <typename T, int CIN>
__global__ __launch_bounds__(256) void ifnet_stage_input_kernel(const float* __restrict__ i0, const float* __restrict__ i1,
                                                                const float* __restrict__ flow, const float* __restrict__ mask, int H, int W,
                                                                float timestep, float inv_sf, float fmul, int hs, int ws, T* dst, int Cpad) {
    __shared__ __attribute__((aligned(16))) T tile[64 * 64];   // [pixel of the unshuffled map][Cpad <= 64]
    constexpr int CX = CIN == 12 ? 8 : 7;
    const int wo = ws / 2;
    const long n = (long)(hs / 2) * wo;
    const int pl = threadIdx.x >> 2, sub = threadIdx.x & 3;
    for (long base = (long)blockIdx.x * 64; base < n; base += (long)gridDim.x * 64) {   // uniform: barriers inside
        reinterpret_cast<uint4*>(tile)[threadIdx.x] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4*>(tile)[threadIdx.x + 256] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        const long i = base + pl;
        if (i < n) {
            const int yo = (int)(i / wo), xo = (int)(i - (long)yo * wo);
            const int y = 2 * yo + (sub >> 1), x = 2 * xo + (sub & 1);
            int y0, y1, x0, x1;
            float wy, wx;
            bilin_setup(y, inv_sf, H, &y0, &y1, &wy);
            bilin_setup(x, inv_sf, W, &x0, &x1, &wx);
            const bool hx = wx != 0.f, hy = wy != 0.f;
            const IfnX a00 = ifnet_x_at<CIN>(i0, i1, flow, mask, H, W, timestep, y0, x0);
            const IfnX a01 = hx ? ifnet_x_at<CIN>(i0, i1, flow, mask, H, W, timestep, y0, x1) : a00;
            const IfnX a10 = hy ? ifnet_x_at<CIN>(i0, i1, flow, mask, H, W, timestep, y1, x0) : a00;
            const IfnX a11 = (hx && hy) ? ifnet_x_at<CIN>(i0, i1, flow, mask, H, W, timestep, y1, x1) : (hx ? a01 : a10);
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float top = a00.v[c] * (1.f - wx) + a01.v[c] * wx;
                const float bot = a10.v[c] * (1.f - wx) + a11.v[c] * wx;
                tile[pl * Cpad + c * 4 + sub] = cvt<T>((top * (1.f - wy) + bot * wy) * (c < CX ? 1.0f : fmul));
            }
        }
        __syncthreads();
        const int ppp = Cpad / 8;                               // 16-byte pieces per pixel
        for (int k = threadIdx.x; k < 64 * ppp; k += 256)
            if (base + k / ppp < n) *reinterpret_cast<uint4*>(dst + base * Cpad + (long)k * 8) = reinterpret_cast<const uint4*>(tile)[k];
        __syncthreads();
    }
}

// ifnet_accumulate_kernel reading lastconv's output in place, so that the [4h][4w][6] copy is never written: tmp[Y][X][c6] =
// src[(Y >> 2, X >> 2)][pos * 6 + c6], pos = ((Y & 1) * 2 + (X & 1)) * 4 + ((Y >> 1) & 1) * 2 + ((X >> 1) & 1) - depth_to_space4_kernel's
// permutation with the engine's lastconv rows re-ordered (ifnet.hip finalize) so that a tap's 5 values are contiguous (read at
// c6 * 16 + pos, five cache lines per tap, this kernel took 48 us per block at 1080p against 26 + 13 for the two it replaces).
__global__ __launch_bounds__(256) void ifnet_accumulate_d2s_kernel(const float* __restrict__ t96, int hf, int wf, int cs, int H, int W, float scale,
                                                                   float* flow, float* mask, int first) {
    const long n = (long)H * W;
    const float inv_sf = 1.0f / scale;
    const int hs = 4 * hf, ws = 4 * wf;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        int y0, y1, x0, x1;
        float wy, wx;
        bilin_setup(y, inv_sf, hs, &y0, &y1, &wy);
        bilin_setup(x, inv_sf, ws, &x0, &x1, &wx);
        auto at = [&](int Y, int X) {
            return t96 + ((size_t)(Y >> 2) * wf + (X >> 2)) * cs + (((Y & 1) * 2 + (X & 1)) * 4 + ((Y >> 1) & 1) * 2 + ((X >> 1) & 1)) * 6;
        };
        // a tap's five values sit at a multiple of 24 bytes (cs % 2 == 0): two 8-byte loads and one 4-byte load instead of five
        auto tap = [&](int Y, int X, float* o) {
            const float* q = at(Y, X);
            const float2 a = *reinterpret_cast<const float2*>(q), b = *reinterpret_cast<const float2*>(q + 2);
            o[0] = a.x, o[1] = a.y, o[2] = b.x, o[3] = b.y, o[4] = q[4];
        };
        float p00[5], p01[5], p10[5], p11[5];
        tap(y0, x0, p00);
        tap(y0, x1, p01);
        tap(y1, x0, p10);
        tap(y1, x1, p11);
        float v[5];
#pragma unroll
        for (int c = 0; c < 5; ++c)
            v[c] = (p00[c] * (1.f - wx) + p01[c] * wx) * (1.f - wy) + (p10[c] * (1.f - wx) + p11[c] * wx) * wy;
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!first) f = *reinterpret_cast<const float4*>(flow + i * 4);      // one 16-byte load / store per pixel instead of four of 4 bytes
        f.x = f.x + v[0] * scale;
        f.y = f.y + v[1] * scale;
        f.z = f.z + v[2] * scale;
        f.w = f.w + v[3] * scale;
        *reinterpret_cast<float4*>(flow + i * 4) = f;
        mask[i] = (first ? 0.f : mask[i]) + v[4];
    }
}

// merged = warp(I0, flow[:2]) * sigmoid(mask) + warp(I1, flow[2:4]) * (1 - sigmoid(mask)); crop to H x W;
// optional fp32 RGB; uint8 BGR = round_half_even(clamp(x, 0, 1) * 255)
__global__ __launch_bounds__(256) void ifnet_blend_kernel(const float* __restrict__ i0, const float* __restrict__ i1,
                                                          const float* __restrict__ flow, const float* __restrict__ mask,
                                                          int Hp, int Wp, int H, int W, uint8_t* out_bgr, float* out_rgb) {
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        const size_t ip = (size_t)y * Wp + x;
        const float4 f = *reinterpret_cast<const float4*>(flow + ip * 4);
        float a[3], b[3];
        warp_px(i0, Hp, Wp, x + f.x, y + f.y, a);
        warp_px(i1, Hp, Wp, x + f.z, y + f.w, b);
        const float m = 1.0f / (1.0f + expf(-mask[ip]));
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = a[c] * m + b[c] * (1.f - m);
            if (out_rgb) out_rgb[i * 3 + c] = v;
            if (out_bgr) out_bgr[i * 3 + (2 - c)] = (uint8_t)rintf(fminf(fmaxf(v, 0.f), 1.f) * 255.f);
        }
    }
}

static int grid_for(long n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }

}  // namespace fw


// ---- motion-blur reduction (interpolation.py:403-455): Pillow's UnsharpMask on uint8 H x W x C images ---------------------
// One pass of libImaging's extended box blur along x (stride 1 pixel) or y, edge-replicated, 24-bit fixed point:
//   out = (ww * sum_{k=-r..r} in[clamp(i + k)] + fw * (in[clamp(i - r - 1)] + in[clamp(i + r + 1)]) + 2^23) >> 24
// The image is tiny next to the nets (6 passes over a frame): one thread per sample, taps read straight from global memory
// (the 2r + 3 taps of neighbouring threads share cache lines; r = 1 for the reference's radius 2).
__global__ void box_blur_pass_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, int C, int vertical,
                                        int r, unsigned ww, unsigned fw) {
    const long n = (long)H * W * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long px = i / C;
        const int x = (int)(px % W), y = (int)(px / W);
        const int pos = vertical ? y : x, len = vertical ? H : W;
        const long stride = vertical ? (long)W * C : C;
        const uint8_t* line = src + (vertical ? (long)x * C + c : (long)y * W * C + c);
        unsigned acc = 0;
        for (int k = -r; k <= r; ++k) {
            const int q = min(max(pos + k, 0), len - 1);
            acc += line[q * stride];
        }
        const unsigned far = (unsigned)line[min(max(pos - r - 1, 0), len - 1) * stride] + (unsigned)line[min(max(pos + r + 1, 0), len - 1) * stride];
        dst[i] = (uint8_t)((acc * ww + far * fw + (1u << 23)) >> 24);
    }
}

// d = in - blur; |d| > threshold: clip8(in + d * percent / 100) (C division, truncating), else in     (UnsharpMask.c)
__global__ void unsharp_finish_u8_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ blur, long n, int percent,
                                         int threshold, uint8_t* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int a = src[i], d = a - (int)blur[i];
        int v = a;
        if (abs(d) > threshold) v = min(max(a + d * percent / 100, 0), 255);
        out[i] = (uint8_t)v;
    }
}


// ---- launchers used by the whole-model engine (ifnet.hip) -------------------------------------------------------------------
namespace fw {
static int ifn_grid(long n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }
void launch_ifnet_u8_to_rgb(const uint8_t* in_bgr, int H, int W, int Hp, int Wp, float* out, hipStream_t st) {
    // (four pixels per thread with 16-byte stores was slower: 14.3 against 10.7 us per 1080p frame)
    hipLaunchKernelGGL(u8_to_rgb_f32_kernel, dim3(ifn_grid((long)Hp * Wp)), dim3(256), 0, st, in_bgr, H, W, Hp, Wp, out);
}
void launch_resize_bilinear(const float* src, int Hs, int Ws, int C, float* dst, int Hd, int Wd, int dst_cstride, int dst_coff,
                            float scale_factor, float mul, hipStream_t st) {
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(ifn_grid((long)Hd * Wd)), dim3(256), 0, st, src, Hs, Ws, C, dst, Hd, Wd,
                       dst_cstride, dst_coff, 1.0f / scale_factor, mul);
}
void launch_ifnet_build_x(const float* i0, const float* i1, const float* flow, const float* mask, int H, int W, float timestep,
                          float* x, hipStream_t st) {
    hipLaunchKernelGGL(ifnet_build_x_kernel, dim3(ifn_grid((long)H * W)), dim3(256), 0, st, i0, i1, flow, mask, H, W, timestep, x);
}
void launch_unshuffle2_cast(DType dt, const void* src, bool src_f32, int h, int w, int C, int src_cstride, void* dst,
                            int dst_channels, hipStream_t st) {
    dim3 g(ifn_grid((long)(h / 2) * (w / 2) * dst_channels)), b(256);
    if (!src_f32 && !(C & 1) && !(src_cstride & 1) && !(dst_channels & 7) && !((size_t)src & 3) && !((size_t)dst & 15)) {
        const dim3 g8(ifn_grid((long)(h / 2) * (w / 2) * (dst_channels / 8)));
        if (dt == DT_BF16)
            hipLaunchKernelGGL((unshuffle_typed8_kernel<__bf16>), g8, b, 0, st, (const __bf16*)src, h, w, C, src_cstride, (__bf16*)dst, dst_channels);
        else
            hipLaunchKernelGGL((unshuffle_typed8_kernel<_Float16>), g8, b, 0, st, (const _Float16*)src, h, w, C, src_cstride, (_Float16*)dst, dst_channels);
        return;
    }
    if (dt == DT_BF16) {
        if (src_f32)
            hipLaunchKernelGGL((unshuffle_cast_kernel<__bf16, float>), g, b, 0, st, (const float*)src, h, w, C, src_cstride, (__bf16*)dst, dst_channels);
        else
            hipLaunchKernelGGL((unshuffle_cast_kernel<__bf16, __bf16>), g, b, 0, st, (const __bf16*)src, h, w, C, src_cstride, (__bf16*)dst, dst_channels);
    } else {
        if (src_f32)
            hipLaunchKernelGGL((unshuffle_cast_kernel<_Float16, float>), g, b, 0, st, (const float*)src, h, w, C, src_cstride, (_Float16*)dst, dst_channels);
        else
            hipLaunchKernelGGL((unshuffle_cast_kernel<_Float16, _Float16>), g, b, 0, st, (const _Float16*)src, h, w, C, src_cstride, (_Float16*)dst, dst_channels);
    }
}
void launch_depth_to_space4(const float* src, int h, int w, int cs, float* dst, hipStream_t st) {
    hipLaunchKernelGGL(depth_to_space4_kernel, dim3(ifn_grid((long)h * w * 96)), dim3(256), 0, st, src, h, w, cs, dst);
}
void launch_ifnet_accumulate(const float* tmp, int hs, int ws, int H, int W, float scale, float* flow, float* mask, int first,
                             hipStream_t st) {
    hipLaunchKernelGGL(ifnet_accumulate_kernel, dim3(ifn_grid((long)H * W)), dim3(256), 0, st, tmp, hs, ws, H, W, scale, flow, mask, first);
}
void launch_ifnet_stage_input(DType dt, const float* i0, const float* i1, const float* flow, const float* mask, int H, int W, float timestep,
                              int s, void* dst, int dst_channels, hipStream_t st) {
    if (dst_channels > 64 || (dst_channels & 7) || H % (2 * s) || W % (2 * s)) throw Error(1, "ifnet stage input: bad shape");
    const int hs = H / s, ws = W / s;
    const long blocks = ((long)(hs / 2) * (ws / 2) + 63) / 64;
    const dim3 g((unsigned)(blocks < 8192 ? blocks : 8192)), b(256);
    const float inv_sf = 1.0f / (1.0f / s), fmul = 1.0f / s;
    if (flow && !mask) throw Error(1, "ifnet stage input: flow without mask");
#define FW_SI(T, CIN) hipLaunchKernelGGL((ifnet_stage_input_kernel<T, CIN>), g, b, 0, st, i0, i1, flow, mask, H, W, timestep, inv_sf, fmul, hs, ws, (T*)dst, dst_channels)
    if (dt == DT_BF16) { if (flow) FW_SI(__bf16, 12); else FW_SI(__bf16, 7); }
    else { if (flow) FW_SI(_Float16, 12); else FW_SI(_Float16, 7); }
#undef FW_SI
}
void launch_ifnet_accumulate_d2s(const float* t96, int hf, int wf, int cs, int H, int W, float scale, float* flow, float* mask, int first, hipStream_t st) {
    if ((cs & 1) || ((size_t)t96 & 7) || ((size_t)flow & 15)) throw Error(1, "ifnet accumulate: t96 must be 8-byte aligned with an even channel stride, flow 16-byte aligned");
    hipLaunchKernelGGL(ifnet_accumulate_d2s_kernel, dim3(ifn_grid((long)H * W)), dim3(256), 0, st, t96, hf, wf, cs, H, W, scale, flow, mask, first);
}
void launch_ifnet_blend(const float* i0, const float* i1, const float* flow, const float* mask, int Hp, int Wp, int H, int W,
                        uint8_t* out_bgr, float* out_rgb, hipStream_t st) {
    hipLaunchKernelGGL(ifnet_blend_kernel, dim3(ifn_grid((long)H * W)), dim3(256), 0, st, i0, i1, flow, mask, Hp, Wp, H, W, out_bgr, out_rgb);
}
}  // namespace fw

using namespace fw;

namespace {
int fail(int code, const std::string& m) {
    fw::last_error_ref() = m;
    return code;
}
#define FW_LAUNCHED()                                                            \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) return fail(FW_ERR_HIP, hipGetErrorString(e_));    \
    } while (0)
}  // namespace

extern "C" {

int fw_u8_to_rgb_f32(const uint8_t* in_bgr, int height, int width, int padded_height, int padded_width, float* out,
                     void* stream) {
    if (!in_bgr || !out || height < 1 || width < 1 || padded_height < height || padded_width < width)
        return fail(FW_ERR_INVALID, "fw_u8_to_rgb_f32: bad argument");
    fw::launch_ifnet_u8_to_rgb(in_bgr, height, width, padded_height, padded_width, out, (hipStream_t)stream);
    FW_LAUNCHED();
    return FW_OK;
}

int fw_resize_bilinear_f32(const float* src, int src_h, int src_w, int channels, float* dst, int dst_h, int dst_w,
                           int dst_cstride, int dst_coff, float scale_factor, float mul, void* stream) {
    if (!src || !dst || src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1 || channels < 1 || dst_cstride < dst_coff + channels ||
        !(scale_factor > 0.f))
        return fail(FW_ERR_INVALID, "fw_resize_bilinear_f32: bad argument");
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for((long)dst_h * dst_w)), dim3(256), 0, (hipStream_t)stream, src,
                       src_h, src_w, channels, dst, dst_h, dst_w, dst_cstride, dst_coff, 1.0f / scale_factor, mul);
    FW_LAUNCHED();
    return FW_OK;
}

int fw_ifnet_build_x(const float* img0, const float* img1, const float* flow, const float* mask, int height, int width,
                     float timestep, float* x, void* stream) {
    if (!img0 || !img1 || !x || height < 1 || width < 1 || ((flow == nullptr) != (mask == nullptr)))
        return fail(FW_ERR_INVALID, "fw_ifnet_build_x: bad argument");
    hipLaunchKernelGGL(ifnet_build_x_kernel, dim3(grid_for((long)height * width)), dim3(256), 0, (hipStream_t)stream, img0,
                       img1, flow, mask, height, width, timestep, x);
    FW_LAUNCHED();
    return FW_OK;
}

int fw_unshuffle2_cast(int dtype, const void* src, int src_is_f32, int height, int width, int channels, int src_cstride,
                       void* dst, int dst_channels, void* stream) {
    if (!src || !dst || height < 2 || width < 2 || (height & 1) || (width & 1) || channels < 1 || src_cstride < channels ||
        dst_channels < 4 * channels || (dtype != FW_DTYPE_BF16 && dtype != FW_DTYPE_F16))
        return fail(FW_ERR_INVALID, "fw_unshuffle2_cast: bad argument");
    const long n = (long)(height / 2) * (width / 2) * dst_channels;
    dim3 g(grid_for(n)), b(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FW_DTYPE_BF16) {
        if (src_is_f32)
            hipLaunchKernelGGL((unshuffle_cast_kernel<__bf16, float>), g, b, 0, st, (const float*)src, height, width, channels,
                               src_cstride, (__bf16*)dst, dst_channels);
        else
            hipLaunchKernelGGL((unshuffle_cast_kernel<__bf16, __bf16>), g, b, 0, st, (const __bf16*)src, height, width,
                               channels, src_cstride, (__bf16*)dst, dst_channels);
    } else {
        if (src_is_f32)
            hipLaunchKernelGGL((unshuffle_cast_kernel<_Float16, float>), g, b, 0, st, (const float*)src, height, width,
                               channels, src_cstride, (_Float16*)dst, dst_channels);
        else
            hipLaunchKernelGGL((unshuffle_cast_kernel<_Float16, _Float16>), g, b, 0, st, (const _Float16*)src, height, width,
                               channels, src_cstride, (_Float16*)dst, dst_channels);
    }
    FW_LAUNCHED();
    return FW_OK;
}

int fw_depth_to_space4_f32(const float* src, int height, int width, int src_cstride, float* dst, void* stream) {
    if (!src || !dst || height < 1 || width < 1 || src_cstride < 96) return fail(FW_ERR_INVALID, "fw_depth_to_space4_f32: bad argument");
    hipLaunchKernelGGL(depth_to_space4_kernel, dim3(grid_for((long)height * width * 96)), dim3(256), 0, (hipStream_t)stream,
                       src, height, width, src_cstride, dst);
    FW_LAUNCHED();
    return FW_OK;
}

int fw_ifnet_accumulate(const float* tmp, int tmp_h, int tmp_w, int height, int width, float scale, float* flow, float* mask,
                        int first, void* stream) {
    if (!tmp || !flow || !mask || tmp_h < 1 || tmp_w < 1 || height < 1 || width < 1 || !(scale >= 1.f))
        return fail(FW_ERR_INVALID, "fw_ifnet_accumulate: bad argument");
    hipLaunchKernelGGL(ifnet_accumulate_kernel, dim3(grid_for((long)height * width)), dim3(256), 0, (hipStream_t)stream, tmp,
                       tmp_h, tmp_w, height, width, scale, flow, mask, first);
    FW_LAUNCHED();
    return FW_OK;
}

int fw_ifnet_blend(const float* img0, const float* img1, const float* flow, const float* mask, int padded_height,
                   int padded_width, int height, int width, uint8_t* out_bgr, float* out_rgb_f32, void* stream) {
    if (!img0 || !img1 || !flow || !mask || (!out_bgr && !out_rgb_f32) || height < 1 || width < 1 || padded_height < height ||
        padded_width < width)
        return fail(FW_ERR_INVALID, "fw_ifnet_blend: bad argument");
    hipLaunchKernelGGL(ifnet_blend_kernel, dim3(grid_for((long)height * width)), dim3(256), 0, (hipStream_t)stream, img0, img1,
                       flow, mask, padded_height, padded_width, height, width, out_bgr, out_rgb_f32);
    FW_LAUNCHED();
    return FW_OK;
}

int fw_unsharp_mask_u8(const uint8_t* src, int height, int width, int channels, int box_radius, unsigned ww, unsigned fw_weight,
                       int passes, int percent, int threshold, uint8_t* scratch_a, uint8_t* scratch_b, uint8_t* out, void* stream) {
    if (!src || !scratch_a || !scratch_b || !out || height < 1 || width < 1 || channels < 1 || channels > 4 || box_radius < 0 ||
        box_radius > 1024 || passes < 1 || passes > 8 || percent < 0 || threshold < 0 ||
        (unsigned long long)ww * (2ull * box_radius + 1) + 2ull * fw_weight > (1ull << 24))
        return fail(FW_ERR_INVALID, "fw_unsharp_mask_u8: bad argument");
    const long n = (long)height * width * channels;
    hipStream_t st = (hipStream_t)stream;
    const uint8_t* cur = src;
    uint8_t* bufs[2] = {scratch_a, scratch_b};
    int which = 0;
    for (int axis = 0; axis < 2; ++axis)
        for (int k = 0; k < passes; ++k) {
            hipLaunchKernelGGL(box_blur_pass_u8_kernel, dim3(grid_for(n)), dim3(256), 0, st, cur, bufs[which], height, width, channels,
                               axis, box_radius, ww, fw_weight);
            FW_LAUNCHED();
            cur = bufs[which];
            which ^= 1;
        }
    hipLaunchKernelGGL(unsharp_finish_u8_kernel, dim3(grid_for(n)), dim3(256), 0, st, src, cur, n, percent, threshold, out);
    FW_LAUNCHED();
    return FW_OK;
}

}  // extern "C"
