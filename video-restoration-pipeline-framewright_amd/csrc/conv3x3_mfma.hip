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
constexpr int ROW_PIECES = HALO_W * 4;                    // 136 16-byte pieces per halo row
constexpr int ACT_PIECES = HALO_H * ROW_PIECES;           // 2448 pieces per 32-channel chunk
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

// One global_load_lds_dwordx4: every active lane copies 16 bytes from (base + voff) to LDS[lds_dst + 16 * lane];
// base and lds_dst are wave-uniform.  Inline asm so that hipcc does not count these loads: with the builtin it
// drains them (s_waitcnt vmcnt(0)) before the first ds_read of the chunk being computed, which serialises the
// pipeline (cdna_hip_programming.md §5 "Three .s-level traps" (b)).  The matching wait is the explicit vmcnt(0)
// at the top of the chunk loop.  M0 is written and restored inside the statement (§5.7).
__device__ __forceinline__ void glds16(const void* base, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(base), "s"(lds_dst)
        : "memory");
}

// Fragments of one (k-step, dx) group: the wave's 6 halo rows at column offset dx and the 3 (dy) weight
// fragments per cout tile -> 12*CT MFMAs.
template <int CT>
struct Frags {
    uint4 x[6];
    uint4 w[3][CT];
};

template <typename T, int CT, int EPI>
__global__ __launch_bounds__(256, 1) void conv3x3_mfma_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) uint4 lds[Smem<CT>::TOTAL];
    constexpr int BUF = Smem<CT>::BUF;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r = lane & 31;
    const int h = lane >> 5;

    // ---- persistent workgroup: a contiguous range of tiles ---------------------------------------------------
    // Blocks b and b+8 share an XCD (and its L2): logical id lb puts the blocks of one XCD on a contiguous band of
    // tiles, so vertically neighbouring tiles (shared halo rows) and the weights are served by one L2.
    const int NB = gridDim.x;
    const int xcd = blockIdx.x & 7;
    const int qn = NB >> 3, rn = NB & 7;
    const int lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int tiles_x = (p.W + TILE_W - 1) / TILE_W;
    const int ntiles = tiles_x * ((p.H + TILE_H - 1) / TILE_H);
    const int t_lo = (int)((long)lb * ntiles / NB);
    const int t_hi = (int)((long)(lb + 1) * ntiles / NB);
    if (t_lo >= t_hi) return;

    // ---- tile-invariant per-lane DMA plan -----------------------------------------------------------------------
    const int ups = p.upsample2x;
    const int Ws = ups ? (p.W >> 1) : p.W;
    const long org = ((long)Ws + 1) * p.in_cstride;  // elements between the DMA base and the tile origin
    unsigned rel[ACT_ITERS];                          // byte offset of this lane's piece from the DMA base
    int rp[ACT_ITERS];                                // (halo row << 8) | halo px, or -1 for the unused tail
#pragma unroll
    for (int i = 0; i < ACT_ITERS; ++i) {
        const int idx = (wave + 4 * i) * 64 + lane;
        const int row = idx / ROW_PIECES;
        const int rm = idx - row * ROW_PIECES;
        const int px = rm >> 2;
        const int s = (rm & 3) ^ ((px >> 2) & 3);  // which 8-channel slot lands at this LDS position
        const int srow = ups ? ((row - 1) >> 1) : (row - 1);
        const int spx = ups ? ((px - 1) >> 1) : (px - 1);
        rel[i] = (unsigned)(((long)srow * Ws + spx) * p.in_cstride + s * 8 + org) * 2u;
        rp[i] = (idx < ACT_PIECES) ? ((row << 8) | px) : -1;
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
    const char* in_b = reinterpret_cast<const char*>(p.in);
    const char* w_b = reinterpret_cast<const char*>(p.wpk);

    // DMA of (tile origin ty0/tx0, chunk c) into pipeline stage `st`; positions outside the image are zeroed.
    auto issue = [&](int ty0, int tx0, int c, int st) {
        const int sy0 = ups ? (ty0 >> 1) : ty0;
        const int sx0 = ups ? (tx0 >> 1) : tx0;
        const char* base = in_b + (((long)sy0 * Ws + sx0) * p.in_cstride - org + c * p.in_pstride) * 2;
        const unsigned dst = lds_base + (unsigned)(st * BUF) * 16u;
#pragma unroll
        for (int i = 0; i < ACT_ITERS; ++i) {
            const int j = wave + 4 * i;
            if (j < ACT_INSTR) {
                const int gy = ty0 - 1 + (rp[i] >> 8);
                const int gx = tx0 - 1 + (rp[i] & 255);
                const bool ok = rp[i] >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                if (ok)
                    glds16(base, rel[i], dst + (unsigned)j * 1024u);
                else if (rp[i] >= 0)
                    lds[st * BUF + j * 64 + lane] = make_uint4(0, 0, 0, 0);
            }
        }
        const char* wc = w_b + (size_t)c * (W_FRAGS * CT * 1024);
#pragma unroll
        for (int i = 0; i < (W_FRAGS * CT + 3) / 4; ++i) {
            const int f = wave + 4 * i;
            if (f < W_FRAGS * CT) glds16(wc + f * 1024, (unsigned)lane * 16u, dst + (unsigned)(ACT_REGION + f * 64) * 16u);
        }
    };

    // ---- fragment-read plan ------------------------------------------------------------------------------------
    int rd_off[3][2];  // [dx][ks]: piece index of (halo row 4*wave, px r+dx, k-step ks) for this lane
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            rd_off[dx][ks] = (4 * wave) * ROW_PIECES + (r + dx) * 4 + ((2 * ks + h) ^ (((r + dx) >> 2) & 3));

    auto load_group = [&](Frags<CT>& f, const uint4* a, const uint4* wl, int g) {
        const int ks = g / 3, dx = g - ks * 3;
#pragma unroll
        for (int row = 0; row < 6; ++row) f.x[row] = a[row * ROW_PIECES + rd_off[dx][ks]];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) f.w[dy][ct] = wl[(((dy * 3 + dx) * 2 + ks) * CT + ct) * 64];
    };

    f32x16 acc[4][CT];
    auto mfma_group = [&](const Frags<CT>& f) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int row = 0; row < 4; ++row)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[row][ct] = Op<T>::mfma(f.w[dy][ct], f.x[row + dy], acc[row][ct]);
    };

    f32x16 bias_v[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) bias_v[ct][i] = p.bias[32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h];

    const int nch = p.cin_chunks;
    int st = 0;
    issue((t_lo / tiles_x) * TILE_H, (t_lo % tiles_x) * TILE_W, 0, 0);

    for (int t = t_lo; t < t_hi; ++t) {
        const int y0 = (t / tiles_x) * TILE_H;
        const int x0 = (t % tiles_x) * TILE_W;
#pragma unroll
        for (int row = 0; row < 4; ++row)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[row][ct] = bias_v[ct];

        for (int c = 0; c < nch; ++c) {
            // this chunk has landed (each wave waits for its own DMAs, then the barrier) and every wave is done
            // reading the other stage, which is refilled next
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (c + 1 < nch)
                issue(y0, x0, c + 1, st ^ 1);
            else if (t + 1 < t_hi)
                issue(((t + 1) / tiles_x) * TILE_H, ((t + 1) % tiles_x) * TILE_W, 0, st ^ 1);

            const uint4* a = lds + st * BUF;
            const uint4* wl = a + ACT_REGION + lane;
            Frags<CT> fa, fb;  // register double buffer: the next group's ds_reads fly under this group's MFMAs
            load_group(fa, a, wl, 0);
            load_group(fb, a, wl, 1);
            mfma_group(fa);
            load_group(fa, a, wl, 2);
            mfma_group(fb);
            load_group(fb, a, wl, 3);
            mfma_group(fa);
            load_group(fa, a, wl, 4);
            mfma_group(fb);
            load_group(fb, a, wl, 5);
            mfma_group(fa);
            mfma_group(fb);
            st ^= 1;
        }

        // ---- epilogue (the next tile's first chunk is already in flight) ----------------------------------------
        const int x = x0 + r;
#pragma unroll
        for (int row = 0; row < 4; ++row) {
            const int y = y0 + 4 * wave + row;
            if (y >= p.H || x >= p.W) continue;
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
                const size_t pix = (size_t)y * p.W + x;
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
                            T* dst = reinterpret_cast<T*>(p.out) + pix * p.out_cstride + p.out_coff + ct * p.out_pstride + 8 * g + 4 * h;
                            *reinterpret_cast<uint2*>(dst) = Op<T>::pack4(o[0], o[1], o[2], o[3]);
                        }
                    }
                }
            }
        }
    }
}

static int num_cus() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        return v;
    }();
    return n;
}

template <typename T>
static void launch_typed(int cout_tiles, ConvEpilogue epi, const ConvParams& p, hipStream_t stream) {
    const int tiles = ((p.W + TILE_W - 1) / TILE_W) * ((p.H + TILE_H - 1) / TILE_H);
    // persistent workgroups: one per CU (LDS-limited), each walks a contiguous range of tiles
    dim3 grid(tiles < num_cus() ? tiles : num_cus()), block(256);
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
    if (p.in_cstride < 32 || (p.in_cstride & 7) || p.in_pstride < 32 || (p.in_pstride & 7) ||
        (p.in_pstride == 32 && p.in_cstride < 32 * p.cin_chunks))
        throw Error(1, "conv3x3: bad input channel/plane stride");
    if (p.out && cout_tiles == 2 && (p.out_pstride < 32 || (p.out_pstride & 3)))
        throw Error(1, "conv3x3: bad output plane stride");
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
