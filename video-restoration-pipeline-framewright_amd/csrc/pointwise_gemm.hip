// 1x1 convolution with many channels as a pipelined MFMA GEMM (the deep levels of NAFNet: 256 / 512 / 1024 channels).
//
// nn_ops.hip's pointwise kernel stages one 32-channel chunk at a time through registers with two barriers per chunk - fine
// where the 1x1 convs are HBM-bound (width 64 / 128, full and half resolution), but at the deep levels, where a NAFBlock is
// ~100 GFLOP of GEMM on a few thousand pixels, it ran at 237 TFLOP/s (profiles/r02_tap_kernel_stats.csv: 28 deep blocks were
// half of the forward).  This kernel is the conv3x3 MFMA machinery with one tap:
//
//   D[cout][pixel] += W[cout][k] * A[pixel][k],  v_mfma_f32_16x16x32, pixel on the lane (as in conv3x3_mfma.hip)
//   * workgroup tile 256 pixels x 256 output channels, 8 waves: wave w owns pixels [64 (w & 3), +64) and output-channel
//     tiles [8 (w >> 2), +8) - 32 accumulator tiles (128 registers); a B fragment feeds 8 MFMAs, an A fragment 4;
//   * K in blocks of 64 channels: per block 32 KiB of activations (two 32-channel chunk images [pixel][4 x 16 B], the
//     XOR swizzle of conv_common.h) and 32 KiB of weight fragments arrive by LDS-DMA (global_load_lds_dwordx4, batches of
//     four KiB per wave behind one M0 write), one block ahead, double buffered (128 KiB of LDS);
//   * persistent workgroups, one per CU, each walking a contiguous range of (pixel tile, channel tile) pairs as one flattened
//     (tile, K block) pipeline: the next tile's first block is in flight while the epilogue stores;
//   * epilogues of the NAFBlock (nn_ops.hip PointwiseMode): typed store, SimpleGate (the two halves of the gate are packed
//     into one workgroup's channel tiles by the host), beta / gamma residual into the fp32 stream.
// The SCA scale of conv3 (a per-input-channel factor known only at run time) is folded into a scaled copy of the packed
// weights by a small kernel per forward (pw16_scale_weights_kernel) instead of touching the activations.
#include "fw_internal.h"
#include "conv_common.h"

namespace fw {

constexpr int G_PX = 256;                       // pixels per workgroup tile
constexpr int G_N = 256;                        // (virtual) output channels per workgroup tile
constexpr int G_KB = 64;                        // channels per K block
constexpr int G_A_PIECES = G_PX * 4 * 2;        // 2048 pieces: two chunk images
constexpr int G_W_PIECES = 2 * (G_N / 16) * 64; // 2048 pieces: 2 chunks x 16 channel tiles x 1 KiB
constexpr int G_STAGE = G_A_PIECES + G_W_PIECES;
static_assert(2 * G_STAGE * 16 <= 160 * 1024, "LDS");

// NH = 2: a workgroup tile is 256 pixels x 128 output channels (one half of a packed 256-channel tile; a wave owns 4 channel tiles
// instead of 8) - for launches whose 256 x 256 tiles would leave half of the CUs idle (NAFNet's middle level: 8160 pixels x 1024
// channels = 128 tiles).  Not for PW_GATE (its x1 / x2 pairing lives inside a wave's 8 tiles).
template <typename T, int MODE, int NH = 1>
__global__ __launch_bounds__(512, 2) void pw_gemm_kernel(const PointwiseParams p) {
    constexpr int NJ = 8 / NH;                  // 16-channel tiles per wave
    static_assert(NH == 1 || MODE != PW_GATE, "half tiles: not for the gate");
    __shared__ __attribute__((aligned(16))) uint4 lds[2 * G_STAGE];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, q = lane & 15, sl = lane >> 4;
    const int pg = wave & 3, ch = wave >> 2;   // pixel group (64 px), channel half (8 tiles of 16)

    const int NB = gridDim.x;
    const int xcd = blockIdx.x & 7;
    const int qn = NB >> 3, rn = NB & 7;
    const int lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int n_tiles = p.N_tiles / 8 * NH;                  // 256- (NH = 2: 128-) channel tiles (N_tiles counts 32-channel tiles)
    const long m_tiles = (p.M + G_PX - 1) / G_PX;
    const long ntiles = m_tiles * n_tiles;                   // t = mt * n_tiles + nt: a workgroup's consecutive tiles share A rows
    const long t_lo = lb * ntiles / NB, t_hi = (long)(lb + 1) * ntiles / NB;
    if (t_lo >= t_hi) return;
    const int nkb = p.K / G_KB;
    const long nitems = (t_hi - t_lo) * nkb;

    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
    const char* a_b = reinterpret_cast<const char*>(p.a);
    const char* w_b = reinterpret_cast<const char*>(p.wpk16);
    const unsigned lane16 = lane * 16;
    const long row_bytes = p.lda * 2;

    // ---- DMA stream: item (f_t, f_kb) next; wave w issues A pieces 4w .. 4w+3 (chunk w >> 2, pixel groups 4 (w & 3) + i) and
    //      weight KiB 4w .. 4w+3 of the block ---------------------------------------------------------------------------------
    long f_t = t_lo;
    int f_kb = 0;
    unsigned voff[4];
    auto plan_rows = [&](long mt) {   // per-lane source offsets of this wave's four A pieces in pixel tile mt (rows clamped to M - 1)
        const long m0 = mt * G_PX;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pxl = 16 * (4 * pg + i) + (lane >> 2);
            long m = m0 + pxl;
            if (m > p.M - 1) m = p.M - 1;
            const int s = (lane & 3) ^ halo_swz(pxl);
            voff[i] = (unsigned)((m - m0) * row_bytes + s * 16) + (unsigned)(3 - i) * 1024u;
        }
    };
    auto issue = [&](int stage) {
        const long mt = f_t / n_tiles;
        const int nt = (int)(f_t - mt * n_tiles);
        if (f_kb == 0) plan_rows(mt);
        const char* a_src = a_b + mt * G_PX * row_bytes + ((long)f_kb * G_KB + 32 * ch) * 2;
        const unsigned a_dst = (unsigned)(stage * G_STAGE + ch * (G_PX * 4) + (4 * pg + 3) * 64);
        glds16_batch_a4(a_src, voff, lds_base + a_dst * 16u);
        if constexpr (NH == 1) {
            const char* w_src = w_b + ((size_t)nt * nkb + f_kb) * (G_W_PIECES * 16) + (4 * wave + 4) * 1024;
            glds16_batch_w<4>(w_src, lane16, lds_base + (unsigned)(stage * G_STAGE + G_A_PIECES + (4 * wave + 4) * 64) * 16u);
        } else {
            // half tile: per chunk the 8 KiB of channel tiles 8 half .. 8 half + 7; wave w: chunk w >> 2, tiles 2 (w & 3), + 1 -> LDS [chunk][8 tiles]
            const int wc = wave >> 2, wt = 2 * (wave & 3);
            const char* w_src = w_b + (((size_t)(nt >> 1) * nkb + f_kb) * 2 + wc) * (16 * 1024) + ((nt & 1) * 8 + wt) * 1024;
            const unsigned w_dst = lds_base + (unsigned)(stage * G_STAGE + G_A_PIECES + (wc * 8 + wt) * 64) * 16u;
            glds16(w_src, lane16, w_dst);
            glds16(w_src + 1024, lane16, w_dst + 1024u);
        }
        if (++f_kb == nkb) {
            f_kb = 0;
            ++f_t;
        }
    };

    // B-fragment read offsets: pixel 64 pg + 16 t + q, slot sl of chunk image c
    int rd_b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int px = 64 * pg + 16 * t + q;
        rd_b[t] = px * 4 + (sl ^ halo_swz(px));
    }

    f32x4 acc[4][NJ];
    issue(0);
    long n = 0;
    for (long t = t_lo; t < t_hi; ++t) {
        const long mt = t / n_tiles;
        const int nt = (int)(t - mt * n_tiles);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kb = 0; kb < nkb; ++kb, ++n) {
            // item n has landed (each wave waits for its own DMAs, then the barrier); a tile's first item after an epilogue was
            // waited for ahead of the stores (vmcnt counts stores too)
            if (kb > 0 || t == t_lo) FW_WAIT_VMCNT(0);
            __syncthreads();
            if (n + 1 < nitems) issue((int)((n + 1) & 1));
            const uint4* st = lds + (n & 1) * G_STAGE;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint4* ai = st + c * (G_PX * 4);
                const uint4* wi = st + G_A_PIECES + (c * (16 / NH) + NJ * ch) * 64 + lane;
                uint4 xb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xb[i] = ai[rd_b[i]];
                uint4 wf[3];
                wf[0] = wi[0];
                wf[1] = wi[64];
                FW_SB();
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (j + 2 < NJ) wf[(j + 2) % 3] = wi[(j + 2) * 64];
                    FW_SB();
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = Op<T>::mfma16(wf[j % 3], xb[i], acc[i][j]);
                    FW_SB();
                }
            }
        }
        // the next tile's first block has been in flight for an item: wait for it ahead of the stores
        FW_WAIT_VMCNT(0);

        // ---- epilogue: lane holds pixel 64 pg + 16 i + q, channels 16 (8 ch + j) + 4 sl + {0..3} of the tile's 256 ------------
        const long m0 = mt * G_PX + 64 * pg + q;
        if constexpr (MODE == PW_RESIDUAL) {
            // y = res + (acc + bias) * chan_scale into the fp32 stream
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nn = (256 / NH) * nt + 16 * (NJ * ch + j) + 4 * sl;
                const f32x4 bs = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nn) : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 cs = *reinterpret_cast<const f32x4*>(p.chan_scale + nn);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const long m = m0 + 16 * i;
                    if (m < p.M) {
                        const f32x4 rs = *reinterpret_cast<const f32x4*>(p.res_f32 + m * p.ldf + nn);
                        *reinterpret_cast<f32x4*>(p.out_f32 + m * p.ldf + nn) = rs + (acc[i][j] + bs) * cs;
                    }
                }
            }
        } else {
            // typed output, 16 bytes per lane: v_permlane16_swap pairs the fragments of two neighbouring 16-channel tiles so that
            // a lane owns one whole 8-channel slot of its pixel (conv3x3_mfma.hip's store).  PW_GATE: the wave's tiles 0-3 are
            // x1, tiles 4-7 the matching x2 channels (pack_pointwise_weights16 lays the rows out that way): out = x1 * x2.
            constexpr int NOUT = MODE == PW_GATE ? 4 : NJ;       // output tiles of 16 channels per wave
            const int ls = (sl & 1) ? 2 + (sl >> 1) : (sl >> 1);
            const long n_half = (long)p.N_tiles * 16;            // PW_GATE: bias of x2 sits N / 2 behind x1's
            const int n_wave = MODE == PW_GATE ? 128 * nt + 64 * ch : (256 / NH) * nt + (128 / NH) * ch;   // first output channel of this wave
#pragma unroll
            for (int jp = 0; jp < NOUT / 2; ++jp) {
                f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, b2 = b0, b3 = b0;
                if (p.bias) {
                    b0 = *reinterpret_cast<const f32x4*>(p.bias + n_wave + 32 * jp + 4 * sl);
                    b1 = *reinterpret_cast<const f32x4*>(p.bias + n_wave + 32 * jp + 16 + 4 * sl);
                    if constexpr (MODE == PW_GATE) {
                        b2 = *reinterpret_cast<const f32x4*>(p.bias + n_half + n_wave + 32 * jp + 4 * sl);
                        b3 = *reinterpret_cast<const f32x4*>(p.bias + n_half + n_wave + 32 * jp + 16 + 4 * sl);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 oa = acc[i][2 * jp] + b0, ob = acc[i][2 * jp + 1] + b1;
                    if constexpr (MODE == PW_GATE) {
                        oa = oa * (acc[i][4 + 2 * jp] + b2);
                        ob = ob * (acc[i][4 + 2 * jp + 1] + b3);
                    }
                    const uint2 pa = Op<T>::pack4(oa[0], oa[1], oa[2], oa[3]);
                    const uint2 pb = Op<T>::pack4(ob[0], ob[1], ob[2], ob[3]);
                    const u32x2 sx = __builtin_amdgcn_permlane16_swap(pa.x, pb.x, false, false);
                    const u32x2 sy = __builtin_amdgcn_permlane16_swap(pa.y, pb.y, false, false);
                    const long m = m0 + 16 * i;
                    if (m < p.M)
                        store16(reinterpret_cast<char*>(p.out_typed) + (m * p.ldo + n_wave + 32 * jp + ls * 8) * 2,
                                make_uint4(sx[0], sy[0], sx[1], sy[1]));
                }
            }
        }
    }
}

// dst = packed weights scaled per input channel: element j of fragment lane l of chunk c multiplies channel 32 c + 8 (l >> 4) + j
template <typename T>
__global__ __launch_bounds__(256) void pw16_scale_weights_kernel(const uint4* __restrict__ src, const float* __restrict__ scale, long frags,
                                                                 int nkb, uint4* __restrict__ dst) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < frags * 64; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long f = i >> 6;                       // fragment index: ((nt * nkb + kb) * 2 + c) * 16 + tile
        const int c = (int)((f >> 4) & 1);
        const int kb = (int)((f >> 5) % nkb);
        const int k0 = kb * G_KB + 32 * c + 8 * (lane >> 4);
        const uint4 v = src[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        unsigned o[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float lo, hi;
            if constexpr (sizeof(T) == 2 && __is_same(T, __bf16)) {
                lo = __builtin_bit_cast(float, w[h] << 16);
                hi = __builtin_bit_cast(float, w[h] & 0xffff0000u);
            } else {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 pr = __builtin_bit_cast(h2, w[h]);
                lo = (float)pr[0];
                hi = (float)pr[1];
            }
            lo *= scale[k0 + 2 * h];
            hi *= scale[k0 + 2 * h + 1];
            typedef T t2 __attribute__((ext_vector_type(2)));
            const t2 r = {(T)lo, (T)hi};
            o[h] = __builtin_bit_cast(unsigned, r);
        }
        dst[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

static int gemm_cus() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        return v;
    }();
    return n;
}

bool pointwise_gemm_eligible(const PointwiseParams& p) {
    return p.wpk16 && !p.a_f32 && !p.gather2x2 && !p.ln_w && !p.a_scale && p.K >= 128 && (p.K % G_KB) == 0 && (p.N_tiles % 8) == 0 &&
           (p.lda % 8) == 0 && (p.mode == PW_STORE || p.mode == PW_GATE || p.mode == PW_RESIDUAL) &&
           (p.mode == PW_RESIDUAL ? p.out_f32 && p.res_f32 && p.chan_scale && (p.ldf % 4) == 0 : p.out_typed && !p.out_f32 && (p.ldo % 8) == 0);
}

void launch_pointwise_gemm(DType dt, const PointwiseParams& p, hipStream_t st) {
    if (!pointwise_gemm_eligible(p)) throw Error(1, "pointwise gemm: shape not eligible");
    long tiles = ((p.M + G_PX - 1) / G_PX) * (p.N_tiles / 8);
    static const bool half_on = [] {   // FW_PW_GEMM_HALF=0: always 256-channel tiles (A/B)
        const char* e = getenv("FW_PW_GEMM_HALF");
        return !e || atoi(e) != 0;
    }();
    const bool half = half_on && p.mode != PW_GATE && 2 * tiles <= gemm_cus();   // 256 x 256 tiles would leave half of the CUs idle
    if (half) tiles *= 2;
    dim3 grid((unsigned)(tiles < gemm_cus() ? tiles : gemm_cus())), block(512);
#define FW_G(MODE)                                                                              \
    do {                                                                                        \
        if (dt == DT_BF16)                                                                      \
            hipLaunchKernelGGL((pw_gemm_kernel<__bf16, MODE>), grid, block, 0, st, p);          \
        else                                                                                    \
            hipLaunchKernelGGL((pw_gemm_kernel<_Float16, MODE>), grid, block, 0, st, p);        \
    } while (0)
#define FW_GH(MODE)                                                                             \
    do {                                                                                        \
        if (dt == DT_BF16)                                                                      \
            hipLaunchKernelGGL((pw_gemm_kernel<__bf16, MODE, 2>), grid, block, 0, st, p);       \
        else                                                                                    \
            hipLaunchKernelGGL((pw_gemm_kernel<_Float16, MODE, 2>), grid, block, 0, st, p);     \
    } while (0)
    if (half && p.mode == PW_STORE) FW_GH(PW_STORE);
    else if (half) FW_GH(PW_RESIDUAL);
    else if (p.mode == PW_STORE) FW_G(PW_STORE);
    else if (p.mode == PW_GATE) FW_G(PW_GATE);
    else FW_G(PW_RESIDUAL);
#undef FW_G
#undef FW_GH
    FW_HIP_CHECK(hipGetLastError());
}

void launch_pw16_scale_weights(DType dt, const void* src, const float* scale, int N, int K, void* dst, hipStream_t st) {
    const long frags = (long)(N / 16) * (K / 32);
    const int blocks = (int)((frags * 64 + 255) / 256 < 2048 ? (frags * 64 + 255) / 256 : 2048);
    if (dt == DT_BF16)
        hipLaunchKernelGGL((pw16_scale_weights_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, (const uint4*)src, scale, frags, K / G_KB, (uint4*)dst);
    else
        hipLaunchKernelGGL((pw16_scale_weights_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, (const uint4*)src, scale, frags, K / G_KB, (uint4*)dst);
    FW_HIP_CHECK(hipGetLastError());
}

// Host-side packer: w[cout][K] fp32 -> [n tile of 256][K block of 64][chunk of 32][16-channel tile (16)][lane][8], the A operand
// of v_mfma_f32_16x16x32 (row = lane & 15, k = 8 (lane >> 4) + j).  gate: the workgroup tile holds 128 gated outputs - per
// channel half (64 outputs) tiles 0-3 are x1 rows, tiles 4-7 the matching x2 rows (x2 = x1 + cout / 2).  cout % 256 == 0.
size_t pack_pointwise_weights16(DType dt, const float* w, int cout, int K, int gate, uint16_t* dst) {
    const int ntl = cout / G_N, nkb = K / G_KB;
    const size_t n = (size_t)ntl * nkb * 2 * 16 * 64 * 8;
    if (!dst) return n;
    size_t o = 0;
    for (int nt = 0; nt < ntl; ++nt)
        for (int kb = 0; kb < nkb; ++kb)
            for (int c = 0; c < 2; ++c)
                for (int tile = 0; tile < 16; ++tile)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            int co;
                            if (gate) {
                                const int half = tile >> 3, jj = tile & 7;
                                co = (jj < 4 ? 0 : cout / 2) + 128 * nt + 64 * half + 16 * (jj & 3) + (lane & 15);
                            } else {
                                co = 256 * nt + 16 * tile + (lane & 15);
                            }
                            const int k = G_KB * kb + 32 * c + 8 * (lane >> 4) + j;
                            dst[o++] = f32_to_operand(dt, w[(size_t)co * K + k]);
                        }
    return n;
}

}  // namespace fw
