// conv3x3 (stride 1, zero pad 1) as an implicit GEMM on the CDNA4 matrix cores.
//
// This is kernel K1/K2/K3/K4 of SURVEY.md §8a: every 3x3 convolution of the RRDBNet that the
// reference runs through third-party basicsr (call site reference
// src/framewright/processors/pytorch_realesrgan.py:107-127,223) and whose residual-dense arithmetic is
// spelled out in-tree at src/framewright/processors/aesrgan_face.py:171-204 (ResidualDenseBlock / RRDB)
// and :249-269 (trunk + nearest-x2 upsample tail).
//
// GEMM orientation (chosen for NHWC stores, not translated from any CUDA tiling):
//     D[cout][pixel] += W[cout][k] * X[k][pixel]        k = (tap, cin)
//   * A operand  = weights, pre-packed on the host into 1-KiB MFMA fragments (one coalesced
//                  global_load_dwordx4 per wave per fragment, L2 resident, no LDS needed);
//   * B operand  = activations: a 32-pixel row segment x 16 input channels, read from an LDS halo tile
//                  with ds_read_b128 (8 consecutive channels of one pixel = 16 bytes per lane);
//   * D          = v_mfma_f32_32x32x16 accumulators: the PIXEL is on the lane, 4 consecutive output
//                  channels sit in 4 consecutive registers -> 8-byte NHWC stores straight from registers.
//
// Work decomposition: one 256-thread workgroup (4 waves, one per SIMD) owns a 16x32-pixel output tile; wave w
// owns rows 4w..4w+3.  K is walked in 32-channel chunks; per chunk the 18x34-pixel halo tile (39 KiB) is
// staged global -> registers -> LDS with an XOR swizzle that makes the ds_read_b128 fragment reads
// bank-conflict free (see lds_slot()).
#include "fw_internal.h"

namespace fw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Op;
template <>
struct Op<__bf16> {
    static __device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                       0, 0, 0);
    }
    static __device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
        bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
        return __builtin_bit_cast(uint2, v);
    }
};
template <>
struct Op<_Float16> {
    static __device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    }
    static __device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
        f16x4 v = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
        return __builtin_bit_cast(uint2, v);
    }
};

constexpr int TILE_H = 16;
constexpr int TILE_W = 32;
constexpr int HALO_H = TILE_H + 2;                // 18
constexpr int HALO_W = TILE_W + 2;                // 34
constexpr int LDS_PIECES = HALO_H * HALO_W * 4;   // 16-byte pieces per 32-channel chunk = 2448
constexpr int STAGE_ITERS = (LDS_PIECES + 255) / 256;  // 10

// LDS image of one chunk: [halo row][halo px][4 slots of 16 B]; slot s (= 8 channels) of pixel p is stored
// at slot s ^ ((p >> 2) & 3).  A fragment read takes, for 16 lanes with distinct p mod 16, the 256-byte
// bank row positions (p & 3) * 64 + (s ^ ((p >> 2) & 3)) * 16: all 16 distinct -> conflict free.
__device__ __forceinline__ int lds_slot(int row, int px, int s) { return (row * HALO_W + px) * 4 + (s ^ ((px >> 2) & 3)); }

template <typename T, int CT, int EPI>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(const ConvParams p) {
    __shared__ uint4 lds[LDS_PIECES];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int r = lane & 31;
    const int h = lane >> 5;

    const int tiles_x = (p.W + TILE_W - 1) / TILE_W;
    const int tile_y = blockIdx.x / tiles_x;
    const int tile_x = blockIdx.x - tile_y * tiles_x;
    const int y0 = tile_y * TILE_H;
    const int x0 = tile_x * TILE_W;

    // ---- per-thread staging plan (chunk invariant) --------------------------------------------------
    const int ups = p.upsample2x;
    const int Ws = ups ? (p.W >> 1) : p.W;
    long src_off[STAGE_ITERS];  // element offset of the 8-channel piece inside chunk 0, or -1
    int dst_idx[STAGE_ITERS];
#pragma unroll
    for (int i = 0; i < STAGE_ITERS; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / (HALO_W * 4);
        const int rem = idx - row * (HALO_W * 4);
        const int px = rem >> 2;
        const int s = rem & 3;
        const int gy = y0 - 1 + row;
        const int gx = x0 - 1 + px;
        const bool ok = (idx < LDS_PIECES) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const int sy = ups ? (gy >> 1) : gy;
        const int sx = ups ? (gx >> 1) : gx;
        src_off[i] = ok ? ((long)sy * Ws + sx) * p.in_cstride + s * 8 : -1;
        dst_idx[i] = (idx < LDS_PIECES) ? lds_slot(row, px, s) : -1;
    }

    // ---- accumulators, initialised with the bias ------------------------------------------------------
    f32x16 acc[4][CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x16 b;
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = p.bias[32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h];
#pragma unroll
        for (int row = 0; row < 4; ++row) acc[row][ct] = b;
    }

    const T* in = reinterpret_cast<const T*>(p.in);
    const uint4* wpk = reinterpret_cast<const uint4*>(p.wpk) + lane;

    // fragment-read plan: pixel r+dx of halo row (4*wave + row + dy)
    int rd_base[3];
    int rd_swz[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        rd_base[dx] = (r + dx) * 4;
        rd_swz[dx] = ((r + dx) >> 2) & 3;
    }

    for (int c = 0; c < p.cin_chunks; ++c) {
        // stage chunk c: global -> registers
        uint4 v[STAGE_ITERS];
#pragma unroll
        for (int i = 0; i < STAGE_ITERS; ++i) {
            v[i] = make_uint4(0, 0, 0, 0);
            if (src_off[i] >= 0) v[i] = *reinterpret_cast<const uint4*>(in + src_off[i] + c * 32);
        }
        __syncthreads();  // everyone is done reading the previous chunk
#pragma unroll
        for (int i = 0; i < STAGE_ITERS; ++i)
            if (dst_idx[i] >= 0) lds[dst_idx[i]] = v[i];
        __syncthreads();

        const uint4* wc = wpk + (size_t)c * (9 * 2 * CT * 64);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3;
            const int dx = t - dy * 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 wf[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wf[ct] = wc[((t * 2 + ks) * CT + ct) * 64];
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    const uint4 xf =
                        lds[(4 * wave + row + dy) * (HALO_W * 4) + rd_base[dx] + ((2 * ks + h) ^ rd_swz[dx])];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[row][ct] = Op<T>::mfma(wf[ct], xf, acc[row][ct]);
                }
            }
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------
    const int x = x0 + r;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const int y = y0 + 4 * wave + row;
        if (y >= p.H || x >= p.W) continue;
        const size_t pix = (size_t)y * p.W + x;
        if constexpr (EPI == EPI_IMAGE) {
            if (h == 0 && y < p.img_H && x < p.img_W) {
                const size_t pix = (size_t)y * p.img_W + x;
                const float cr = acc[row][0][0], cg = acc[row][0][1], cb = acc[row][0][2];
                if (p.out_rgb) {
                    float* o = p.out_rgb + pix * 3;
                    o[0] = cr;
                    o[1] = cg;
                    o[2] = cb;
                }
                if (p.out_u8) {
                    uint8_t* o = p.out_u8 + pix * 3;
                    o[0] = (uint8_t)rintf(fminf(fmaxf(cb, 0.f), 1.f) * 255.f);
                    o[1] = (uint8_t)rintf(fminf(fmaxf(cg, 0.f), 1.f) * 255.f);
                    o[2] = (uint8_t)rintf(fminf(fmaxf(cr, 0.f), 1.f) * 255.f);
                }
            }
        } else {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 32 * ct + 8 * g + 4 * h;
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = acc[row][ct][4 * g + j];
                    if constexpr (EPI == EPI_RESIDUAL) {
                        const f32x4 r1 = *reinterpret_cast<const f32x4*>(p.res1 + pix * (32 * CT) + c0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = o[j] * p.s1 + r1[j];
                        if (p.res2) {
                            const f32x4 r2 = *reinterpret_cast<const f32x4*>(p.res2 + pix * (32 * CT) + c0);
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = o[j] * p.s2 + r2[j];
                        }
                    } else {
                        if (p.act) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.2f * o[j]);
                        }
                    }
                    if (p.out_f32) {
                        f32x4 of = {o[0], o[1], o[2], o[3]};
                        *reinterpret_cast<f32x4*>(p.out_f32 + pix * (32 * CT) + c0) = of;
                    }
                    if (p.out) {
                        T* dst = reinterpret_cast<T*>(p.out) + pix * p.out_cstride + p.out_coff + c0;
                        *reinterpret_cast<uint2*>(dst) = Op<T>::pack4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
        }
    }
}

template <typename T>
static void launch_typed(int cout_tiles, ConvEpilogue epi, const ConvParams& p, hipStream_t stream) {
    const int tiles = ((p.W + TILE_W - 1) / TILE_W) * ((p.H + TILE_H - 1) / TILE_H);
    dim3 grid(tiles), block(256);
    if (cout_tiles == 1 && epi == EPI_STORE)
        hipLaunchKernelGGL((conv3x3_mfma_kernel<T, 1, EPI_STORE>), grid, block, 0, stream, p);
    else if (cout_tiles == 2 && epi == EPI_STORE)
        hipLaunchKernelGGL((conv3x3_mfma_kernel<T, 2, EPI_STORE>), grid, block, 0, stream, p);
    else if (cout_tiles == 2 && epi == EPI_RESIDUAL)
        hipLaunchKernelGGL((conv3x3_mfma_kernel<T, 2, EPI_RESIDUAL>), grid, block, 0, stream, p);
    else if (cout_tiles == 1 && epi == EPI_IMAGE)
        hipLaunchKernelGGL((conv3x3_mfma_kernel<T, 1, EPI_IMAGE>), grid, block, 0, stream, p);
    else
        throw Error(1, "conv3x3: unsupported (cout_tiles, epilogue) combination");
    FW_HIP_CHECK(hipGetLastError());
}

void launch_conv3x3(DType dt, int cout_tiles, ConvEpilogue epi, const ConvParams& p, hipStream_t stream) {
    if (p.H <= 0 || p.W <= 0 || p.cin_chunks <= 0) throw Error(1, "conv3x3: empty problem");
    if (p.upsample2x && ((p.H | p.W) & 1)) throw Error(1, "conv3x3: upsample2x needs even output size");
    if (p.in_cstride < 32 * p.cin_chunks || (p.in_cstride & 7)) throw Error(1, "conv3x3: bad input channel stride");
    if (p.out && ((p.out_cstride & 3) || (p.out_coff & 3))) throw Error(1, "conv3x3: output slice must be 8-byte aligned");
    if (dt == DT_BF16)
        launch_typed<__bf16>(cout_tiles, epi, p, stream);
    else
        launch_typed<_Float16>(cout_tiles, epi, p, stream);
}

// ---- host-side packing ----------------------------------------------------------------------------------
uint16_t f32_to_operand(DType dt, float f) {
    if (dt == DT_BF16) {
        uint32_t u;
        memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
        u += 0x7fffu + ((u >> 16) & 1u);                                            // round to nearest even
        return (uint16_t)(u >> 16);
    }
    _Float16 hv = (_Float16)f;
    uint16_t o;
    memcpy(&o, &hv, 2);
    return o;
}

float operand_to_f32(DType dt, uint16_t v) {
    if (dt == DT_BF16) {
        uint32_t u = (uint32_t)v << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
    _Float16 hv;
    memcpy(&hv, &v, 2);
    return (float)hv;
}

// Fragment order: [chunk c][tap t = ky*3+kx][ks][cout tile ct][lane][j], value =
//   w[cout = 32*ct + (lane & 31)][cin = 32*c + 16*ks + 8*(lane >> 5) + j][ky][kx]      (zero outside)
// which is exactly the v_mfma_f32_32x32x16 A-operand map (row = lane & 31, k = 8*(lane >> 5) + j).
size_t pack_conv3x3_weights(DType dt, const float* w, int cout, int cin, int cout_tiles, int cin_chunks,
                            uint16_t* dst) {
    const size_t n = (size_t)cin_chunks * 9 * 2 * cout_tiles * 64 * 8;
    if (!dst) return n;
    size_t o = 0;
    for (int c = 0; c < cin_chunks; ++c)
        for (int t = 0; t < 9; ++t)
            for (int ks = 0; ks < 2; ++ks)
                for (int ct = 0; ct < cout_tiles; ++ct)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int co = 32 * ct + (lane & 31);
                            const int ci = 32 * c + 16 * ks + 8 * (lane >> 5) + j;
                            float val = 0.f;
                            if (co < cout && ci < cin) val = w[((size_t)co * cin + ci) * 9 + t];
                            dst[o++] = f32_to_operand(dt, val);
                        }
    return n;
}

}  // namespace fw
