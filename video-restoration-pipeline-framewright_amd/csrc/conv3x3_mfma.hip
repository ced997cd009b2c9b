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
// owns rows 4w..4w+3.  K is walked in 32-channel chunks; per chunk the 18x34-pixel halo tile (39 KiB) and the
// chunk's weight fragments (18 KiB per 32 output channels) are staged global -> LDS by LDS-DMA
// (global_load_lds_dwordx4) one chunk ahead, with an XOR swizzle that makes the ds_read_b128 fragment reads
// bank-conflict free; both the halo tile and the weight fragments arrive by LDS-DMA, double buffered.
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
constexpr int HALO_H = TILE_H + 2;                        // 18
constexpr int HALO_W = TILE_W + 2;                        // 34
constexpr int ACT_PIECES = HALO_H * HALO_W * 4;           // 16-byte pieces per 32-channel chunk = 2448
constexpr int ACT_INSTR = (ACT_PIECES + 63) / 64;         // 39 wave-instructions of 1 KiB
constexpr int ACT_REGION = ACT_INSTR * 64;                // 2496 pieces (tail of the last instruction unused)
constexpr int ACT_ITERS = (ACT_INSTR + 3) / 4;            // 10 per wave
constexpr int W_FRAGS = 18;                               // 9 taps x 2 k-steps of 16, per cout tile

// LDS image of one activation chunk: [halo row][halo px][4 slots of 16 B]; slot s (= 8 channels) of pixel p is
// stored at slot s ^ ((p >> 2) & 3).  A fragment read takes, for 16 lanes with distinct p mod 16, the 256-byte
// bank row positions (p & 3) * 64 + (s ^ ((p >> 2) & 3)) * 16: all 16 distinct -> conflict free.
// The image is filled by LDS-DMA (global_load_lds_dwordx4): the LDS destination of a wave-instruction is
// lane-linear, so the swizzle is applied on the per-lane SOURCE address (cdna_hip_programming.md rule 21).

template <int CT>
struct Smem {
    static constexpr int BUF = ACT_REGION + W_FRAGS * CT * 64;  // pieces per pipeline stage
    static constexpr int TOTAL = 2 * BUF;                       // double buffered
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// One global_load_lds_dwordx4: every active lane copies 16 bytes from its own global address to
// LDS[lds_dst + 16 * lane]; lds_dst must be wave-uniform.  M0 is written and restored inside the statement
// (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

template <typename T, int CT, int EPI>
__global__ __launch_bounds__(256, 1) void conv3x3_mfma_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) uint4 lds[Smem<CT>::TOTAL];
    constexpr int BUF = Smem<CT>::BUF;

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int r = lane & 31;
    const int h = lane >> 5;

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so XCD g takes a contiguous band of
    // tiles — neighbouring tiles share halo rows and every tile re-reads the same weights.  Bijective for any
    // tile count (cdna_hip_programming.md §5 "XCD swizzle must be bijective").
    const int ntiles = gridDim.x;
    const int xcd = blockIdx.x & 7;
    const int q = ntiles >> 3, rem = ntiles & 7;
    const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (blockIdx.x >> 3);
    const int tiles_x = (p.W + TILE_W - 1) / TILE_W;
    const int tile_y = tile / tiles_x;
    const int tile_x = tile - tile_y * tiles_x;
    const int y0 = tile_y * TILE_H;
    const int x0 = tile_x * TILE_W;

    // ---- per-lane staging plan (chunk invariant) ---------------------------------------------------------
    const int ups = p.upsample2x;
    const int Ws = ups ? (p.W >> 1) : p.W;
    const T* in = reinterpret_cast<const T*>(p.in);
    const T* src[ACT_ITERS];  // source of this lane's 8-channel piece in chunk 0 (nullptr: outside the image)
#pragma unroll
    for (int i = 0; i < ACT_ITERS; ++i) {
        const int idx = (wave + 4 * i) * 64 + lane;
        const int row = idx / (HALO_W * 4);
        const int rm = idx - row * (HALO_W * 4);
        const int px = rm >> 2;
        const int s = (rm & 3) ^ ((px >> 2) & 3);  // which 8-channel slot lands at this LDS position
        const int gy = y0 - 1 + row;
        const int gx = x0 - 1 + px;
        const bool ok = (idx < ACT_PIECES) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const int sy = ups ? (gy >> 1) : gy;
        const int sx = ups ? (gx >> 1) : gx;
        src[i] = ok ? in + ((size_t)sy * Ws + sx) * p.in_cstride + s * 8 : nullptr;
        // zero padding: the DMA never writes these positions, so clear them once in both stages
        if (!ok && idx < ACT_REGION) {
            lds[idx] = make_uint4(0, 0, 0, 0);
            lds[BUF + idx] = make_uint4(0, 0, 0, 0);
        }
    }
    const uint4* wsrc = reinterpret_cast<const uint4*>(p.wpk) + lane;

    // LDS-DMA issue.  Written as inline asm so that hipcc does not count these loads: with the builtin it drains
    // them (s_waitcnt vmcnt(0)) before the first ds_read of the chunk being computed, which serialises the
    // pipeline (cdna_hip_programming.md §5 "Three .s-level traps" (b)).  The matching wait is the explicit
    // vmcnt(0) at the top of the chunk loop.
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto stage = [&](int c, int buf) {
        const unsigned dst = lds_base + (unsigned)(buf * BUF) * 16u;
#pragma unroll
        for (int i = 0; i < ACT_ITERS; ++i) {
            const int j = wave_u + 4 * i;
            if (j < ACT_INSTR && src[i]) glds16(src[i] + c * 32, dst + (unsigned)j * 1024u);
        }
        const uint4* wc = wsrc + (size_t)c * (W_FRAGS * CT * 64);
#pragma unroll
        for (int i = 0; i < (W_FRAGS * CT + 3) / 4; ++i) {
            const int f = wave_u + 4 * i;
            if (f < W_FRAGS * CT) glds16(wc + f * 64, dst + (unsigned)(ACT_REGION + f * 64) * 16u);
        }
    };

    // ---- accumulators, initialised with the bias -------------------------------------------------------------
    f32x16 acc[4][CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x16 b;
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = p.bias[32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h];
#pragma unroll
        for (int row = 0; row < 4; ++row) acc[row][ct] = b;
    }

    // fragment-read plan: pixel r+dx of halo row (4*wave + row + dy)
    int rd_base[3];
    int rd_swz[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        rd_base[dx] = (4 * wave) * (HALO_W * 4) + (r + dx) * 4;
        rd_swz[dx] = ((r + dx) >> 2) & 3;
    }

    stage(0, 0);
    for (int c = 0; c < p.cin_chunks; ++c) {
        // chunk c has landed (every wave waits for its own DMAs, then the barrier), and every wave is done
        // reading the stage that is about to be refilled
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 1 < p.cin_chunks) stage(c + 1, (c + 1) & 1);

        const uint4* a = lds + (c & 1) * BUF;
        const uint4* wl = a + ACT_REGION + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3;
            const int dx = t - dy * 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 wf[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wf[ct] = wl[((t * 2 + ks) * CT + ct) * 64];
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    const uint4 xf = a[(row + dy) * (HALO_W * 4) + rd_base[dx] + ((2 * ks + h) ^ rd_swz[dx])];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[row][ct] = Op<T>::mfma(wf[ct], xf, acc[row][ct]);
                }
            }
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------
    // The accumulators hold [cout][pixel] with the pixel on the lane; a store straight from registers would touch
    // one cache line per lane.  Each wave transposes its own rows through LDS (row stride padded by 16 B: the
    // ds_write_b128 of 8 consecutive pixels then hits 8 distinct 4-bank groups) and continues in a "pixel-major"
    // layout where 8*CT consecutive lanes own one pixel's channels: residual loads, fp32 trunk stores and typed
    // NHWC stores are then contiguous per pixel.
    __syncthreads();  // every wave is done with the last chunk before LDS is reused
    if constexpr (EPI == EPI_IMAGE) {
        const int x = x0 + r;
#pragma unroll
        for (int row = 0; row < 4; ++row) {
            const int y = y0 + 4 * wave + row;
            if (h == 0 && y < p.img_H && x < p.img_W && y < p.H && x < p.W) {
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
        }
    } else {
        constexpr int NC = 32 * CT;             // output channels
        constexpr int ROWF = NC + 4;            // floats per pixel in the LDS scratch (16-byte pad)
        constexpr int LPP = NC / 4;             // lanes per pixel in the pixel-major layout (8 or 16)
        constexpr int PPI = 64 / LPP;           // pixels per wave-instruction (8 or 4)
        float* scratch = reinterpret_cast<float*>(lds) + wave * (32 * ROWF);
        const int c4 = (lane % LPP) * 4;        // this lane's 4 channels in the pixel-major layout
        const int psub = lane / LPP;
#pragma unroll
        for (int row = 0; row < 4; ++row) {
            const int y = y0 + 4 * wave + row;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[row][ct][4 * g], acc[row][ct][4 * g + 1], acc[row][ct][4 * g + 2],
                               acc[row][ct][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(scratch + r * ROWF + 32 * ct + 8 * g + 4 * h) = v;
                }
            // same wave wrote what it reads: LDS operations of one wave complete in order
#pragma unroll
            for (int it = 0; it < 32 / PPI; ++it) {
                const int px = it * PPI + psub;
                const int x = x0 + px;
                f32x4 o = *reinterpret_cast<const f32x4*>(scratch + px * ROWF + c4);
                if (y < p.H && x < p.W) {
                    const size_t pix = (size_t)y * p.W + x;
                    if constexpr (EPI == EPI_RESIDUAL) {
                        const f32x4 r1 = *reinterpret_cast<const f32x4*>(p.res1 + pix * NC + c4);
                        o = o * p.s1 + r1;
                        if (p.res2) {
                            const f32x4 r2 = *reinterpret_cast<const f32x4*>(p.res2 + pix * NC + c4);
                            o = o * p.s2 + r2;
                        }
                    } else {
                        if (p.act) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.2f * o[j]);
                        }
                    }
                    if (p.out_f32) *reinterpret_cast<f32x4*>(p.out_f32 + pix * NC + c4) = o;
                    if (p.out) {
                        T* dst = reinterpret_cast<T*>(p.out) + pix * p.out_cstride + p.out_coff + c4;
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
