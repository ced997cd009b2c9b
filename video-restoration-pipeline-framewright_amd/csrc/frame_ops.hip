// Frame <-> tensor conversion kernels (K5 of SURVEY.md §8a).  HBM-bound byte work: one thread per output
// pixel, 16-byte vector stores, no LDS.
//
// Follows the pre-processing of RealESRGANer.enhance / pre_process as recorded in SURVEY.md §A.2 and
// called from reference src/framewright/processors/pytorch_realesrgan.py:223: BGR->RGB, /255, HWC->CHW
// (here NHWC), for the x2 model reflect mod-pad to an even size followed by pixel_unshuffle(2)
// (basicsr RRDBNet.forward).
#include "fw_internal.h"

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
template <typename T, int UNSHUFFLE>
__global__ __launch_bounds__(256) void u8_to_nhwc_kernel(const uint8_t* __restrict__ in, int H, int W, T* out,
                                                         int out_cstride, int Ho, int Wo) {
    const long n = (long)Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / Wo);
        const int x = (int)(i - (long)y * Wo);
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.f;
        if constexpr (UNSHUFFLE == 1) {
            const uint8_t* px = in + ((size_t)y * W + x) * 3;
            v[0] = px[2] * (1.f / 255.f);
            v[1] = px[1] * (1.f / 255.f);
            v[2] = px[0] * (1.f / 255.f);
        } else {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int sy = reflect_hi(2 * y + dy, H);
                    const int sx = reflect_hi(2 * x + dx, W);
                    const uint8_t* px = in + ((size_t)sy * W + sx) * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c * 4 + dy * 2 + dx] = px[2 - c] * (1.f / 255.f);
                }
        }
        uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * out_cstride);
        o[0] = pack8<T>(v);
        o[1] = pack8<T>(v + 8);
        o[2] = make_uint4(0, 0, 0, 0);
        o[3] = make_uint4(0, 0, 0, 0);
    }
}

void launch_u8_to_nhwc(DType dt, const uint8_t* in_bgr, int H, int W, void* out, int out_cstride, int unshuffle,
                       hipStream_t stream) {
    if (out_cstride < 32 || (out_cstride & 7)) throw Error(1, "u8_to_nhwc: bad channel stride");
    const int Ho = unshuffle == 2 ? (H + 1) / 2 : H;
    const int Wo = unshuffle == 2 ? (W + 1) / 2 : W;
    if (unshuffle == 2 && (H < 2 || W < 2)) throw Error(1, "u8_to_nhwc: x2 model needs at least 2x2 input");
    const long n = (long)Ho * Wo;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    dim3 grid(blocks), block(256);
#define FW_LAUNCH(T, U) \
    hipLaunchKernelGGL((u8_to_nhwc_kernel<T, U>), grid, block, 0, stream, in_bgr, H, W, (T*)out, out_cstride, Ho, Wo)
    if (dt == DT_BF16) {
        if (unshuffle == 2) FW_LAUNCH(__bf16, 2); else FW_LAUNCH(__bf16, 1);
    } else {
        if (unshuffle == 2) FW_LAUNCH(_Float16, 2); else FW_LAUNCH(_Float16, 1);
    }
#undef FW_LAUNCH
    FW_HIP_CHECK(hipGetLastError());
}

}  // namespace fw
