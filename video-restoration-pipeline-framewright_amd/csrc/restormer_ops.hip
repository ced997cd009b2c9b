// Building blocks of Restormer (the reference's default TAP model: `_load_restormer`,
// src/framewright/processors/tap_denoise.py:299-333, `Restormer(dim=48, num_blocks=[4,6,6,8], num_refinement_blocks=4,
// heads=[1,2,4,8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias')`; the network itself is third-party
// `basicsr.archs.restormer_arch`, absent from the reference tree — architecture per SURVEY.md §A.4, oracle
// oracle/restormer_ref.py, parity unpinned).  Host sequencing: framewright_amd/restormer.py.
//
// Layout: the residual stream is fp32 NHWC with a padded channel stride; everything that feeds an MFMA (1x1 convs through
// pointwise_mfma_kernel, 3x3 convs through conv3x3_mfma_kernel) is operand-typed with channels padded to 32.
//   layernorm_nhwc       per-pixel LayerNorm over the first C channels ('WithBias': (x-mu)/sqrt(var+eps)*w+b), typed out
//   dwconv3x3_nhwc       depthwise 3x3 (no bias); mode 1 fuses the GDFN gate gelu(x1)*x2
//   attn_gram / finish   MDTA "transposed" attention: per head the c x c Gram matrix of L2-normalised q, k over all pixels
//                        (fixed-order two-level reduction -> deterministic), temperature, softmax
//   attn_apply           out[p][c1] = sum_c2 A[head][c1][c2] * v[p][c2]
//   pixel_(un)shuffle2   fp32 NHWC
#include "fw_internal.h"
#include "../../include/framewright_hip.h"

namespace fw {

typedef __bf16 rbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rf16x8 __attribute__((ext_vector_type(8)));
template <typename T> struct V8;
template <> struct V8<__bf16> { using t = rbf16x8; };
template <> struct V8<_Float16> { using t = rf16x8; };

// ---- LayerNorm over channels, fp32 [M][ldx] -> typed [M][ldo]; channels [C, Cz) of the output are zeroed ---------------
// Vector form (C, Cz, ldx, ldo multiples of 4 - every call of the Restormer engine): LPP lanes per pixel, 4 channels per lane
// and pass (16-byte loads, 8-byte stores), 64 / LPP pixels per wave.  The scalar form below (one wave per pixel, a channel
// per lane and pass, 2-byte stores) stays for odd shapes.
template <typename T, int LPP>
__global__ __launch_bounds__(256) void layernorm_nhwc_vec_kernel(const float* __restrict__ x, long ldx, long M, int C,
                                                                 const float* __restrict__ w, const float* __restrict__ b,
                                                                 float eps, T* out, long ldo, int Cz) {
    constexpr int PPW = 64 / LPP;
    constexpr int PASSES = LPP == 64 ? 2 : 1;   // C <= 512
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPP, cl = lane % LPP;
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const long groups = (M + PPW - 1) / PPW;
    for (long gi = wave0; gi < groups; gi += nwaves) {
        const long m = gi * PPW + sub;
        const bool live = m < M;
        const float* row = x + (live ? m : 0) * ldx;
        float v[PASSES][4];
        float s = 0.f;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int c0 = ps * 256 + cl * 4;
            f4 t = {0.f, 0.f, 0.f, 0.f};
            if (c0 < C && live) t = *reinterpret_cast<const f4*>(row + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[ps][j] = t[j];
                s += t[j];
            }
        }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mu = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int c0 = ps * 256 + cl * 4;
            if (c0 < C) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[ps][j] - mu;
                    q += d * d;
                }
            }
        }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);
        if (!live) continue;
        T* orow = out + m * ldo;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int c0 = ps * 256 + cl * 4;
            if (c0 < Cz) {
                T o4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o4[j] = c0 < C ? (T)((v[ps][j] - mu) * rstd * w[c0 + j] + (b ? b[c0 + j] : 0.f)) : (T)0.f;
                uint2 pk;
                __builtin_memcpy(&pk, o4, 8);
                *reinterpret_cast<uint2*>(orow + c0) = pk;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_nhwc_kernel(const float* __restrict__ x, long ldx, long M, int C,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             float eps, T* out, long ldo, int Cz) {
    const int lane = threadIdx.x & 63;
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long m = wave0; m < M; m += nwaves) {
        const float* row = x + m * ldx;
        float v[8];  // C <= 512
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < C ? row[c] : 0.f;
            s += v[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mu = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            const float d = c < C ? v[i] - mu : 0.f;
            q += d * d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);
        T* orow = out + m * ldo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            if (c < C)
                orow[c] = (T)((v[i] - mu) * rstd * w[c] + (b ? b[c] : 0.f));
            else if (c < Cz)
                orow[c] = (T)0.f;
        }
    }
}

// ---- depthwise 3x3, zero padding, typed [H][W][ldx] -> typed [H][W][ldo] -------------------------------------------------
// mode 0: out[c] = dw(x)[c], c < C.   mode 1: out[c] = gelu(dw(x)[c]) * dw(x)[C/2 + c], c < C/2 (exact erf GELU).
// wdw: fp32 [C][9].  C % 8 == 0.
// A thread owns 8 channels of a column of DW_ROWS output pixels: the DW_ROWS + 2 input rows x 3 pixels it needs are 16-byte
// loads issued without a branch between them (addresses clamped into the image, the value masked to zero outside), each
// feeds up to three output rows, and the 72 filter taps of the 8 channels sit in registers for the whole column (the first
// version - one pixel per thread, nine loads each behind its own border test and wait, filters re-read from LDS per pixel -
// ran at a tenth of the memory bound).
constexpr int DW_ROWS = 4;
template <typename T, int MODE>
__global__ __launch_bounds__(256, 2) void dwconv3x3_nhwc_kernel(const T* __restrict__ x, long ldx, int H, int W, int C,
                                                             const float* __restrict__ wdw, T* out, long ldo) {
    extern __shared__ __attribute__((aligned(16))) float dw_w[];  // filters as [tap][C]
    for (int i = threadIdx.x; i < 9 * C; i += 256) {
        const int ch = i / 9, tap = i - ch * 9;
        dw_w[tap * C + ch] = wdw[i];
    }
    __syncthreads();
    const int Co = MODE == 1 ? C / 2 : C;
    const unsigned groups = Co / 8;
    using V = typename V8<T>::t;
    const unsigned row_items = (unsigned)W * groups;
    const int strips = (H + DW_ROWS - 1) / DW_ROWS;
    for (int strip = blockIdx.y; strip < strips; strip += gridDim.y)
        for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < row_items; idx += gridDim.x * blockDim.x) {
            const int xx = (int)(idx / groups);
            const int g = (int)(idx - (unsigned)xx * groups);
            const int y0 = strip * DW_ROWS;
            float res[DW_ROWS][8];
#pragma unroll 1
            for (int half = 0; half < (MODE == 1 ? 2 : 1); ++half) {
                const int c0 = half * Co + g * 8;
                float wr[9][8];  // [tap][channel]
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float4 w0 = *reinterpret_cast<const float4*>(dw_w + t * C + c0);
                    const float4 w1 = *reinterpret_cast<const float4*>(dw_w + t * C + c0 + 4);
                    wr[t][0] = w0.x, wr[t][1] = w0.y, wr[t][2] = w0.z, wr[t][3] = w0.w;
                    wr[t][4] = w1.x, wr[t][5] = w1.y, wr[t][6] = w1.z, wr[t][7] = w1.w;
                }
                float acc[DW_ROWS][8];
#pragma unroll
                for (int o = 0; o < DW_ROWS; ++o)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
#pragma unroll
                for (int r = 0; r < DW_ROWS + 2; ++r) {
                    const int sy = y0 + r - 1;
                    const bool rok = sy >= 0 && sy < H;
                    const int cy = sy < 0 ? 0 : (sy >= H ? H - 1 : sy);
                    V f[3];
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int sx = xx + dx - 1;
                        const int cx = sx < 0 ? 0 : (sx >= W ? W - 1 : sx);
                        f[dx] = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(x + ((long)cy * W + cx) * ldx + c0));
                    }
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int sx = xx + dx - 1;
                        const float m = (rok && sx >= 0 && sx < W) ? 1.f : 0.f;
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = (float)f[dx][j] * m;
#pragma unroll
                        for (int o = 0; o < DW_ROWS; ++o) {
                            const int dy = r - o;  // input row r feeds output row o through tap row dy
                            if (dy >= 0 && dy < 3) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) acc[o][j] += v[j] * wr[dy * 3 + dx][j];
                            }
                        }
                    }
                }
#pragma unroll
                for (int o = 0; o < DW_ROWS; ++o)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
#ifndef FW_DW_GELU_ERFF
                        // erf after A&S 7.1.28 (fw_internal.h), scalar: a third fewer VALU instructions than with erff and no spills; the
                        // packed two-value form needs aligned register pairs and spilled 33 registers here (28.3 against 23.1 us)
                        if (MODE == 1 && half == 0)
                            res[o][j] = gelu_erf1(acc[o][j]);
                        else
#else
                        if (MODE == 1 && half == 0)
                            res[o][j] = 0.5f * acc[o][j] * (1.0f + erff(acc[o][j] * 0.70710678118654752f));
                        else
#endif
                        if (MODE == 1)
                            res[o][j] *= acc[o][j];
                        else
                            res[o][j] = acc[o][j];
                    }
            }
#pragma unroll
            for (int o = 0; o < DW_ROWS; ++o) {
                if (y0 + o >= H) continue;
                V ov;
#pragma unroll
                for (int j = 0; j < 8; ++j) ov[j] = (T)res[o][j];
                *reinterpret_cast<uint4*>(out + ((long)(y0 + o) * W + xx) * ldo + g * 8) = __builtin_bit_cast(uint4, ov);
            }
        }
}

// The same kernel for the many-channel levels (192 / 384 channels on 128^2 / 64^2 pixels: 576 ... 2048 depthwise channels).  There
// every one of the up to 512 blocks of the kernel above staged ALL filters (72 KB at 2048 channels) because its 256 threads spanned
// every channel group.  Here a block owns 32 channel groups (256 output channels: 9 or 18 KB of filters) and walks columns.
template <typename T, int MODE>
__global__ __launch_bounds__(256, 2) void dwconv3x3_nhwc_wide_kernel(const T* __restrict__ x, long ldx, int H, int W, int C,
                                                                  const float* __restrict__ wdw, T* out, long ldo) {
    constexpr int HALVES = MODE == 1 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float dw_w[9 * HALVES * 256];   // [tap][half][256 channels of this block's range]
    const int Co = MODE == 1 ? C / 2 : C;
    const int groups = Co / 8, ngr = (groups + 31) / 32;
    const int gr = blockIdx.x % ngr, cb = blockIdx.x / ngr, ncb = gridDim.x / ngr;
    for (int i = threadIdx.x; i < 9 * HALVES * 256; i += 256) {
        const int tap = i / (HALVES * 256), r = i - tap * (HALVES * 256), half = r >> 8, ch = gr * 256 + (r & 255);
        dw_w[i] = ch < Co ? wdw[(size_t)(half * Co + ch) * 9 + tap] : 0.f;
    }
    __syncthreads();
    const int gl = threadIdx.x & 31, cl = threadIdx.x >> 5;
    const int g = gr * 32 + gl;
    if (g >= groups) return;
    using V = typename V8<T>::t;
    const int strips = (H + DW_ROWS - 1) / DW_ROWS;
    for (int strip = blockIdx.y; strip < strips; strip += gridDim.y)
        for (int xx = cb * 8 + cl; xx < W; xx += ncb * 8) {
            const int y0 = strip * DW_ROWS;
            float res[DW_ROWS][8];
#pragma unroll 1
            for (int half = 0; half < HALVES; ++half) {
                const int c0 = half * Co + g * 8;
                const float* wh = dw_w + half * 256 + gl * 8;
                float wr[9][8];  // [tap][channel]
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float4 w0 = *reinterpret_cast<const float4*>(wh + t * HALVES * 256);
                    const float4 w1 = *reinterpret_cast<const float4*>(wh + t * HALVES * 256 + 4);
                    wr[t][0] = w0.x, wr[t][1] = w0.y, wr[t][2] = w0.z, wr[t][3] = w0.w;
                    wr[t][4] = w1.x, wr[t][5] = w1.y, wr[t][6] = w1.z, wr[t][7] = w1.w;
                }
                float acc[DW_ROWS][8];
#pragma unroll
                for (int o = 0; o < DW_ROWS; ++o)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
#pragma unroll
                for (int r = 0; r < DW_ROWS + 2; ++r) {
                    const int sy = y0 + r - 1;
                    const bool rok = sy >= 0 && sy < H;
                    const int cy = sy < 0 ? 0 : (sy >= H ? H - 1 : sy);
                    V f[3];
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int sx = xx + dx - 1;
                        const int cx = sx < 0 ? 0 : (sx >= W ? W - 1 : sx);
                        f[dx] = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(x + ((long)cy * W + cx) * ldx + c0));
                    }
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int sx = xx + dx - 1;
                        const float m = (rok && sx >= 0 && sx < W) ? 1.f : 0.f;
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = (float)f[dx][j] * m;
#pragma unroll
                        for (int o = 0; o < DW_ROWS; ++o) {
                            const int dy = r - o;
                            if (dy >= 0 && dy < 3) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) acc[o][j] += v[j] * wr[dy * 3 + dx][j];
                            }
                        }
                    }
                }
#pragma unroll
                for (int o = 0; o < DW_ROWS; ++o)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
#ifndef FW_DW_GELU_ERFF
                        // erf after A&S 7.1.28 (fw_internal.h), scalar: a third fewer VALU instructions than with erff and no spills; the
                        // packed two-value form needs aligned register pairs and spilled 33 registers here (28.3 against 23.1 us)
                        if (MODE == 1 && half == 0)
                            res[o][j] = gelu_erf1(acc[o][j]);
                        else
#else
                        if (MODE == 1 && half == 0)
                            res[o][j] = 0.5f * acc[o][j] * (1.0f + erff(acc[o][j] * 0.70710678118654752f));
                        else
#endif
                        if (MODE == 1)
                            res[o][j] *= acc[o][j];
                        else
                            res[o][j] = acc[o][j];
                    }
            }
#pragma unroll
            for (int o = 0; o < DW_ROWS; ++o) {
                if (y0 + o >= H) continue;
                V ov;
#pragma unroll
                for (int j = 0; j < 8; ++j) ov[j] = (T)res[o][j];
                *reinterpret_cast<uint4*>(out + ((long)(y0 + o) * W + xx) * ldo + g * 8) = __builtin_bit_cast(uint4, ov);
            }
        }
}

// ---- MDTA Gram matrices ---------------------------------------------------------------------------------------------------
// qkv typed [M][ld]: q at channel 0, k at k_off; `dim` = heads * ch channels each.  partial[block][dim*ch + 2*dim]:
//   [h][c1][c2] = sum_p q[p][h*ch+c1] * k[p][h*ch+c2] over the block's pixels, then sum q^2 and sum k^2 per channel.
// A thread owns 3x3 patches of (c1, c2): patch index -> (head, c1/3, c2/3); pixels are staged 16 at a time through LDS.
constexpr int GRAM_PX = 16;
constexpr int GRAM_MAX_BLOCKS = 512;
template <typename T>
__global__ __launch_bounds__(256) void attn_gram_kernel(const T* __restrict__ qkv, long ld, long M, int k_off, int heads, int ch,
                                                        float* partial) {
    extern __shared__ __attribute__((aligned(16))) float gq[];  // [GRAM_PX][dim] q then [GRAM_PX][dim] k
    const int dim = heads * ch;
    float* gk = gq + GRAM_PX * dim;
    const int pp = ch / 3;                        // patches per side
    const int npatch = heads * pp * pp;
    constexpr int MAXP = 8;                       // dim*ch/9/256 <= 8 for (384,48), (96,96), (192,48), ...
    float acc[MAXP][9];
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[i][j] = 0.f;
    float nq[2] = {0.f, 0.f}, nk[2] = {0.f, 0.f};  // channels tid and tid + 256 (dim <= 512)
    const long chunks = (M + GRAM_PX - 1) / GRAM_PX;
    for (long cb = blockIdx.x; cb < chunks; cb += gridDim.x) {
        const long p0 = cb * GRAM_PX;
        __syncthreads();
        // staging: 16-byte loads of 8 channels (ld, k_off and dim are multiples of 8)
        {
            using V = typename V8<T>::t;
            const int d8 = dim >> 3;
            for (int i = threadIdx.x; i < GRAM_PX * d8; i += 256) {
                const int px = i / d8, c = (i - px * d8) << 3;
                const long p = p0 + px;
                V vq, vk;
                if (p < M) {
                    vq = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(qkv + p * ld + c));
                    vk = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(qkv + p * ld + k_off + c));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    gq[px * dim + c + j] = p < M ? (float)vq[j] : 0.f;
                    gk[px * dim + c + j] = p < M ? (float)vk[j] : 0.f;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int pt = threadIdx.x + 256 * i;
            if (pt < npatch) {
                const int h = pt / (pp * pp), r = pt - h * pp * pp;
                const int c1 = h * ch + 3 * (r / pp), c2 = h * ch + 3 * (r % pp);
#pragma unroll 4
                for (int px = 0; px < GRAM_PX; ++px) {
                    const float q0 = gq[px * dim + c1], q1 = gq[px * dim + c1 + 1], q2 = gq[px * dim + c1 + 2];
                    const float k0 = gk[px * dim + c2], k1 = gk[px * dim + c2 + 1], k2 = gk[px * dim + c2 + 2];
                    acc[i][0] += q0 * k0; acc[i][1] += q0 * k1; acc[i][2] += q0 * k2;
                    acc[i][3] += q1 * k0; acc[i][4] += q1 * k1; acc[i][5] += q1 * k2;
                    acc[i][6] += q2 * k0; acc[i][7] += q2 * k1; acc[i][8] += q2 * k2;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = threadIdx.x + 256 * i;
            if (c < dim)
                for (int px = 0; px < GRAM_PX; ++px) {
                    nq[i] += gq[px * dim + c] * gq[px * dim + c];
                    nk[i] += gk[px * dim + c] * gk[px * dim + c];
                }
        }
    }
    float* out = partial + (size_t)blockIdx.x * (dim * ch + 2 * dim);
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int pt = threadIdx.x + 256 * i;
        if (pt < npatch) {
            const int h = pt / (pp * pp), r = pt - h * pp * pp;
            const int c1 = 3 * (r / pp), c2 = 3 * (r % pp);
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) out[(h * ch + c1 + a) * ch + c2 + b] = acc[i][a * 3 + b];
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < dim) {
            out[dim * ch + c] = nq[i];
            out[dim * ch + dim + c] = nk[i];
        }
    }
}

// Sum the partials in block order (one thread per element, coalesced across elements; a fixed order -> deterministic) into
// the first row of the workspace.
// ---- the same Gram matrices on the matrix cores ------------------------------------------------------------------------------
// G = q^T k contracts over PIXELS, so the MFMA's k axis must be the pixel axis: lane (row = channel, slot s) of an A / B
// fragment needs 8 consecutive pixels of one channel - a transposed read of the NHWC tensor.  qk_transpose_kernel writes q and k
// once as [pixel group of 8][channel][8 pixels] (16 bytes per (group, channel)); a fragment of the Gram kernel is then one
// 16-byte load per lane, lanes of a 16-lane group on consecutive channels (256 contiguous bytes), no LDS.  Pixels are padded
// with zeros to a multiple of 32 (one v_mfma_f32_16x16x32 step).
template <typename T>
__global__ __launch_bounds__(256) void qk_transpose_kernel(const T* __restrict__ qkv, long ld, long M, long Mp, int k_off, int dim,
                                                           T* qT, T* kT) {
    const int d8 = dim >> 3;
    const long total = (Mp >> 3) * d8 * 2;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % d8);
        const long r = idx / d8;
        const int which = (int)(r & 1);
        const long pg = r >> 1;
        const T* src = qkv + (which ? k_off : 0) + cg * 8;
        unsigned short in[8][8];
#pragma unroll
        for (int px = 0; px < 8; ++px) {
            const long p = pg * 8 + px;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (p < M) v = *reinterpret_cast<const uint4*>(src + p * ld);
            __builtin_memcpy(in[px], &v, 16);
        }
        T* dst = (which ? kT : qT) + (pg * dim + cg * 8) * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            unsigned short o[8];
#pragma unroll
            for (int px = 0; px < 8; ++px) o[px] = in[px][c];
            uint4 v;
            __builtin_memcpy(&v, o, 16);
            *reinterpret_cast<uint4*>(dst + c * 8) = v;
        }
    }
}

typedef float gram_f4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ gram_f4 gram_mfma(uint4 a, uint4 b, gram_f4 c);
template <>
__device__ __forceinline__ gram_f4 gram_mfma<__bf16>(uint4 a, uint4 b, gram_f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(rbf16x8, a), __builtin_bit_cast(rbf16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ gram_f4 gram_mfma<_Float16>(uint4 a, uint4 b, gram_f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rf16x8, a), __builtin_bit_cast(rf16x8, b), c, 0, 0, 0);
}

// grid (blocks, heads).  A block owns a contiguous range of 32-pixel steps, its four waves take every fourth step; a wave
// keeps all NT x NT 16x16 tiles of its head's ch x ch matrix (NT = ch / 16) plus the per-channel sums of squares, the
// waves are then added in a fixed order through LDS and the block writes its row of `partial` ([h][c1][c2], sum q^2, sum k^2 -
// the layout attn_reduce_kernel / attn_finish_kernel read).
template <typename T, int NT>
__global__ __launch_bounds__(256) void attn_gram_mfma_kernel(const T* __restrict__ qT, const T* __restrict__ kT, long Mp, int heads,
                                                             int ldc /* channels per pixel group in qT / kT (>= heads * ch) */, float* partial) {
    constexpr int ch = 16 * NT;
    __shared__ float red[NT * NT * 256 + 2 * NT * 16];
    const int h = blockIdx.y, dim = heads * ch;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, sl = lane >> 4;
    using V = typename V8<T>::t;
    const long steps = Mp >> 5;
    const long s_lo = steps * blockIdx.x / gridDim.x, s_hi = steps * (blockIdx.x + 1) / gridDim.x;
    gram_f4 acc[NT][NT];
    float nq[NT], nk[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        nq[i] = nk[i] = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = gram_f4{0.f, 0.f, 0.f, 0.f};
    }
    for (long st = s_lo + wave; st < s_hi; st += 4) {
        const long pg = st * 4 + sl;
        uint4 a[NT], b[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            a[i] = *reinterpret_cast<const uint4*>(qT + (pg * ldc + h * ch + 16 * i + r) * 8);
            b[i] = *reinterpret_cast<const uint4*>(kT + (pg * ldc + h * ch + 16 * i + r) * 8);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const V qa = __builtin_bit_cast(V, a[i]), kb = __builtin_bit_cast(V, b[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                nq[i] += (float)qa[e] * (float)qa[e];
                nk[i] += (float)kb[e] * (float)kb[e];
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = gram_mfma<T>(a[i], b[j], acc[i][j]);
    }
    // per-channel sums: the four slots of a channel sit in lanes r, r + 16, r + 32, r + 48
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        nq[i] += __shfl_xor(nq[i], 16);
        nq[i] += __shfl_xor(nq[i], 32);
        nk[i] += __shfl_xor(nk[i], 16);
        nk[i] += __shfl_xor(nk[i], 32);
    }
    float* rn = red + NT * NT * 256;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float* d = red + ((i * NT + j) * 64 + lane) * 4 + e;
                        *d = (w ? *d : 0.f) + acc[i][j][e];
                    }
            if (sl == 0) {
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    rn[i * 16 + r] = (w ? rn[i * 16 + r] : 0.f) + nq[i];
                    rn[NT * 16 + i * 16 + r] = (w ? rn[NT * 16 + i * 16 + r] : 0.f) + nk[i];
                }
            }
        }
        __syncthreads();
    }
    // D fragment: lane (r, sl) element e = G[c1 = 16 i + 4 sl + e][c2 = 16 j + r]
    float* out = partial + (size_t)blockIdx.x * ((size_t)dim * ch + 2 * dim);
    for (int t = threadIdx.x; t < NT * NT * 256; t += 256) {
        const int e = t & 3, ln = (t >> 2) & 63, ij = t >> 8;
        const int i = ij / NT, j = ij - i * NT;
        const int c1 = 16 * i + 4 * (ln >> 4) + e, c2 = 16 * j + (ln & 15);
        out[((size_t)h * ch + c1) * ch + c2] = red[t];
    }
    for (int t = threadIdx.x; t < 2 * NT * 16; t += 256) {
        const int which = t / (NT * 16), c = t - which * NT * 16;
        out[(size_t)dim * ch + which * dim + h * ch + c] = rn[t];
    }
}

// Two levels, both in a fixed order (deterministic): thread (e, g) of the first launch sums rows g, g + G, g + 2G, ... into row g
// (read by nobody else: in place), the second launch (G = 1 over the first G rows) sums those into row 0.
__global__ __launch_bounds__(256) void attn_reduce_kernel(float* partial, int nblocks, long stride, int G) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y;
    if (e >= stride || g >= nblocks) return;
    float s = 0.f;
#pragma unroll 8
    for (int b = g; b < nblocks; b += G) s += partial[(size_t)b * stride + e];
    partial[(size_t)g * stride + e] = s;
}

// A[h][c1][:] = softmax_c2(G / (|q_c1| |k_c2|) * temperature[h]) with F.normalize's clamp (norm >= 1e-12), from the reduced
// sums.  One workgroup per (head, c1) row.
__global__ __launch_bounds__(128) void attn_finish_kernel(const float* __restrict__ sums, int rows, long stride, int heads, int ch,
                                                          const float* __restrict__ temperature, float* attn) {
    __shared__ float row[128];
    __shared__ float red[128];
    const int dim = heads * ch;
    const int h = blockIdx.x / ch, c1 = blockIdx.x - h * ch;
    const int c2 = threadIdx.x;
    if (c2 < ch) {
        // the last level of the reduction happens here: rows 0 .. rows-1 of the workspace, in order (deterministic)
        // (loads of eight rows in flight at a time; the adds stay in row order.  Before: one row per L2 round trip, 10.6 us per launch)
        float g = 0.f, kn = 0.f, qn = 0.f;
        const float* s0 = sums + (h * ch + c1) * ch + c2;
        const float* s1 = sums + dim * ch + dim + h * ch + c2;
        const float* s2 = sums + dim * ch + h * ch + c1;
        int b = 0;
        for (; b + 8 <= rows; b += 8) {
            float gv[8], kv[8], qv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                gv[u] = s0[(size_t)(b + u) * stride];
                kv[u] = s1[(size_t)(b + u) * stride];
                qv[u] = s2[(size_t)(b + u) * stride];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) g += gv[u], kn += kv[u], qn += qv[u];
        }
        for (; b < rows; ++b) {
            g += s0[(size_t)b * stride];
            kn += s1[(size_t)b * stride];
            qn += s2[(size_t)b * stride];
        }
        const float dq = fmaxf(sqrtf(qn), 1e-12f), dk = fmaxf(sqrtf(kn), 1e-12f);
        row[c2] = g / (dq * dk) * temperature[h];
    }
    __syncthreads();
    red[c2] = c2 < ch ? row[c2] : -3.0e38f;
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
        if (c2 < o) red[c2] = fmaxf(red[c2], red[c2 + o]);
        __syncthreads();
    }
    const float mx = red[0];
    __syncthreads();
    const float e = c2 < ch ? expf(row[c2] - mx) : 0.f;
    red[c2] = e;
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
        if (c2 < o) red[c2] += red[c2 + o];
        __syncthreads();
    }
    if (c2 < ch) attn[((size_t)h * ch + c1) * ch + c2] = e / red[0];
}

// The attention matrices as ONE block-diagonal [kp x kp] weight in the fragment order of pack_pointwise_weights
// ([chunk][ks][cout tile][lane][8]: cout = 32*tile + (lane & 31), k = 32*chunk + 16*ks + 8*(lane >> 5) + j), operand-typed,
// so that "attn @ v" runs as a 1x1 convolution on the pointwise MFMA GEMM (the zeros of the other heads cost nothing that
// matters next to a VALU mat-vec per pixel).
template <typename T>
__global__ __launch_bounds__(256) void attn_pack_kernel(const float* __restrict__ attn, int heads, int ch, int kp, T* dst) {
    const int dim = heads * ch, nt = kp / 32;
    const long total = (long)nt * 2 * nt * 64 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        long r = i >> 9;
        const int t = (int)(r % nt);
        r /= nt;
        const int ks = (int)(r & 1), c = (int)(r >> 1);
        const int co = 32 * t + (lane & 31), k = 32 * c + 16 * ks + 8 * (lane >> 5) + j;
        float v = 0.f;
        if (co < dim && k < dim && co / ch == k / ch) v = attn[((size_t)(co / ch) * ch + co % ch) * ch + k % ch];
        dst[i] = (T)v;
    }
}

// project_out folded into the attention: y = Wp (A v) = (Wp A) v.  Wc[co][h ch + c2] = sum_c1 Wp[co][h ch + c1] A[h][c1][c2] in the
// fragment order of pack_pointwise_weights (kp x 32 nt), so that "project_out(attn @ v) + x" is ONE 1x1 convolution on v with a
// residual epilogue (before: two GEMM passes over the pixels and a typed tensor between them).
template <typename T>
__global__ __launch_bounds__(256) void attn_proj_pack_kernel(const float* __restrict__ attn, const float* __restrict__ wp, int heads, int ch,
                                                             int kp, int nt, T* dst) {
    const int dim = heads * ch;
    const long total = (long)(kp / 32) * 2 * nt * 64 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        long r = i >> 9;
        const int t = (int)(r % nt);
        r /= nt;
        const int ks = (int)(r & 1), c = (int)(r >> 1);
        const int co = 32 * t + (lane & 31), k = 32 * c + 16 * ks + 8 * (lane >> 5) + j;
        float v = 0.f;
        if (co < dim && k < dim) {
            const int h = k / ch, c2 = k - h * ch;
            const float* w = wp + (size_t)co * dim + h * ch;
            const float* a = attn + (size_t)h * ch * ch + c2;
            // 16 products' loads in flight at a time, summed in c1 order (before: one c1 per L2 round trip, 14.9 us per launch)
            int c1 = 0;
            for (; c1 + 16 <= ch; c1 += 16) {
                float wv[16], av[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = w[c1 + u], av[u] = a[(size_t)(c1 + u) * ch];
#pragma unroll
                for (int u = 0; u < 16; ++u) v += wv[u] * av[u];
            }
            for (; c1 < ch; ++c1) v += w[c1] * a[(size_t)c1 * ch];
        }
        dst[i] = (T)v;
    }
}

// out[p][h*ch + c1] = sum_c2 A[h][c1][c2] * v[p][v_off + h*ch + c2]; typed out, channels [dim, Cz) zeroed.
// thread = (pixel, 8 output channels); A in LDS.
template <typename T>
__global__ __launch_bounds__(256) void attn_apply_kernel(const T* __restrict__ qkv, long ld, long M, int v_off, int heads, int ch,
                                                         const float* __restrict__ attn, T* out, long ldo, int Cz) {
    extern __shared__ __attribute__((aligned(16))) float sa[];  // [dim][ch]
    const int dim = heads * ch;
    for (int i = threadIdx.x; i < dim * ch; i += 256) sa[i] = attn[i];
    __syncthreads();
    const int groups = Cz / 8;
    const long total = M * groups;
    using V = typename V8<T>::t;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long p = idx / groups;
        const int g = (int)(idx - p * groups);
        V o;
        if (g * 8 >= dim) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (T)0.f;
        } else {
            const int h = (g * 8) / ch;             // ch % 8 == 0: a group never straddles heads
            const T* vrow = qkv + p * ld + v_off + h * ch;
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int c2 = 0; c2 < ch; c2 += 8) {
                const V vv = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(vrow + c2));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float* ar = sa + (size_t)(g * 8 + j) * ch + c2;
#pragma unroll
                    for (int t = 0; t < 8; ++t) a[j] += ar[t] * (float)vv[t];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (T)a[j];
        }
        *reinterpret_cast<uint4*>(out + p * ldo + g * 8) = __builtin_bit_cast(uint4, o);
    }
}

// ---- pixel (un)shuffle, fp32 NHWC -------------------------------------------------------------------------------------------
// shuffle:   dst[2y+dy][2x+dx][coff + c] = src[y][x][c*4 + dy*2 + dx],  c < C            (torch.nn.PixelShuffle(2))
// unshuffle: dst[y][x][coff + c*4 + dy*2 + dx] = src[2y+dy][2x+dx][c], c < C            (torch.nn.PixelUnshuffle(2))
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const float* __restrict__ src, long lds_, int H, int W, int C,
                                                             float* dst, long ldd, int coff, int unshuffle) {
    const long total = (long)H * W * C * 4;  // H x W = the LOW-resolution size in both directions
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % (4 * C));
        const long pl = i / (4 * C);
        const int y = (int)(pl / W), x = (int)(pl - (long)y * W);
        const int c = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
        const long hi = ((long)(2 * y + dy) * (2 * W) + (2 * x + dx));
        if (unshuffle)
            dst[pl * ldd + coff + k] = src[hi * lds_ + c];
        else
            dst[hi * ldd + coff + c] = src[pl * lds_ + k];
    }
}

// fp32 [M][ldx] channels [0, C) -> [M][ldd] at coff (concat along channels)
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, long ldx, long M, int C, float* dst,
                                                            long ldd, int coff) {
    const long total = M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        dst[m * ldd + coff + c] = src[m * ldx + c];
    }
}

// ---- AESRGAN AttentionBlock (reference src/framewright/processors/aesrgan_face.py:142-168): full spatial self-attention ----
// P[i][j] = softmax_j(sum_c q[i][c] k[j][c]) over ALL pixels j, operand-typed [M][ldp] with columns [M, ldp) zeroed (ldp = M
// padded to 32: the product P @ v then runs as a 1x1-conv GEMM with K = ldp).  q, k: typed, d <= 8 channels used, 16-byte
// aligned rows.  One wave per row i: three passes over k (max, sum of exp, write) - k is M x 16 bytes, L2-resident.
template <typename T>
__global__ __launch_bounds__(256) void attn_softmax_rows_kernel(const T* __restrict__ q, long ldq, const T* __restrict__ k, long ldk,
                                                                long M, int d, T* P, long ldp) {
    using V = typename V8<T>::t;
    const int lane = threadIdx.x & 63;
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long i = wave0; i < M; i += nwaves) {
        const V qv = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(q + i * ldq));
        float qf[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) qf[c] = c < d ? (float)qv[c] : 0.f;
        auto score = [&](long j) {
            const V kv = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(k + j * ldk));
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) s += qf[c] * (float)kv[c];
            return s;
        };
        float mx = -3.0e38f;
        for (long j = lane; j < M; j += 64) mx = fmaxf(mx, score(j));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (long j = lane; j < M; j += 64) sum += expf(score(j) - mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.0f / sum;
        T* row = P + i * ldp;
        for (long j = lane; j < ldp; j += 64) row[j] = j < M ? (T)(expf(score(j) - mx) * inv) : (T)0.f;
    }
}

// fw_pack_pointwise's fragment order built on the device from a typed matrix held TRANSPOSED: W[co][kk] = src[kk][co]
// (src: typed [K_valid][lds], co < cout; rows kk >= K_valid are zero).  Used for W = v^T in P @ v.
template <typename T>
__global__ __launch_bounds__(256) void pack_pointwise_t_kernel(const T* __restrict__ src, long lds_, long K_valid, int cout, int K_pad,
                                                               T* packed) {
    const int nt = (cout + 31) / 32, chunks = K_pad / 32;
    const long total = (long)chunks * 2 * nt * 64;   // 16-byte pieces
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        long r = idx >> 6;
        const int t = (int)(r % nt);
        r /= nt;
        const int ks = (int)(r & 1);
        const long c = r >> 1;
        const int co = 32 * t + (lane & 31);
        const long k0 = 32 * c + 16 * ks + 8 * (lane >> 5);
        T v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (co < cout && k0 + j < K_valid) ? src[(k0 + j) * lds_ + co] : (T)0.f;
        uint4 o;
        __builtin_memcpy(&o, v, 16);
        reinterpret_cast<uint4*>(packed)[idx] = o;
    }
}

static int blocks_for(long n, int cap) {
    const long b = (n + 255) / 256;
    return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

}  // namespace fw

using namespace fw;

namespace {
int rfail(int code, const std::string& m) {
    fw::last_error_ref() = m;
    return code;
}
template <typename F>
int rguard(F&& f) {
    try {
        f();
        return FW_OK;
    } catch (const fw::Error& e) {
        return rfail(e.code, e.what());
    } catch (const std::exception& e) {
        return rfail(FW_ERR_INTERNAL, e.what());
    }
}
bool bad_dtype(int d) { return d != FW_DTYPE_BF16 && d != FW_DTYPE_F16; }
}  // namespace

extern "C" {

int fw_layernorm_nhwc(int dtype, const float* x, long ldx, long M, int C, const float* weight, const float* bias, float eps,
                      void* out, long ldo, int zero_to, void* stream) {
    if (bad_dtype(dtype) || !x || !weight || !out || M < 1 || C < 1 || C > 512 || zero_to > ldo || C > ldx)
        return rfail(FW_ERR_INVALID, "fw_layernorm_nhwc: bad argument");
    return rguard([&] {
        const int cz = zero_to > C ? zero_to : C;
        if (!(C & 3) && !(cz & 3) && !(ldx & 3) && !(ldo & 3) && !((size_t)x & 15) && !((size_t)out & 7)) {
            const int lanes = (cz + 3) / 4;   // lanes a pixel needs in one pass
            const int lpp = lanes <= 16 ? 16 : (lanes <= 32 ? 32 : 64);
            const int blocks = blocks_for((M + 64 / lpp - 1) / (64 / lpp) * 64, 4096);
#define FW_LNV(T, L) hipLaunchKernelGGL((layernorm_nhwc_vec_kernel<T, L>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, M, C, weight, bias, eps, (T*)out, ldo, cz)
            if (dtype == FW_DTYPE_BF16) {
                if (lpp == 16) FW_LNV(__bf16, 16); else if (lpp == 32) FW_LNV(__bf16, 32); else FW_LNV(__bf16, 64);
            } else {
                if (lpp == 16) FW_LNV(_Float16, 16); else if (lpp == 32) FW_LNV(_Float16, 32); else FW_LNV(_Float16, 64);
            }
#undef FW_LNV
            FW_HIP_CHECK(hipGetLastError());
            return;
        }
        const int blocks = blocks_for(M * 64, 4096);
        if (dtype == FW_DTYPE_BF16)
            hipLaunchKernelGGL((layernorm_nhwc_kernel<__bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, M, C, weight,
                               bias, eps, (__bf16*)out, ldo, zero_to);
        else
            hipLaunchKernelGGL((layernorm_nhwc_kernel<_Float16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, M, C,
                               weight, bias, eps, (_Float16*)out, ldo, zero_to);
        FW_HIP_CHECK(hipGetLastError());
    });
}

size_t fw_pack_pointwise(int dtype, const float* weight, int cout, int k, void* dst) {
    if (bad_dtype(dtype) || cout < 1 || k < 1 || (k & 31)) return 0;
    return pack_pointwise_weights((DType)dtype, weight, cout, k, (uint16_t*)dst);
}

int fw_pointwise_nhwc(int dtype, const void* a, int a_is_f32, long lda, long M, int k, const void* packed_weight,
                      const float* bias, int cout_tiles, void* out_typed, long ldo, float* out_f32, long ldf,
                      const float* res_f32, const float* chan_scale, void* stream) {
    if (bad_dtype(dtype) || !a || !packed_weight || (!out_typed && !out_f32))
        return rfail(FW_ERR_INVALID, "fw_pointwise_nhwc: bad argument");
    if (res_f32 && (!out_f32 || !chan_scale)) return rfail(FW_ERR_INVALID, "fw_pointwise_nhwc: residual needs out_f32 and chan_scale");
    return rguard([&] {
        PointwiseParams p{};
        p.a = a;
        p.a_f32 = a_is_f32;
        p.lda = lda;
        p.M = M;
        p.K = k;
        p.wpk = packed_weight;
        p.bias = bias;
        p.N_tiles = cout_tiles;
        p.mode = res_f32 ? PW_RESIDUAL : PW_STORE;
        p.out_typed = out_typed;
        p.ldo = ldo;
        p.out_f32 = out_f32;
        p.ldf = ldf;
        p.res_f32 = res_f32;
        p.chan_scale = chan_scale;
        launch_pointwise((DType)dtype, p, (hipStream_t)stream);
    });
}

int fw_dwconv3x3_nhwc(int dtype, const void* x, long ldx, int H, int W, int channels, const float* weight, int mode, void* out,
                      long ldo, void* stream) {
    if (bad_dtype(dtype) || !x || !weight || !out || H < 1 || W < 1 || channels < 8 || (channels & 7) || channels > 4096 ||
        (mode != 0 && mode != 1) || (mode == 1 && (channels & 15)))
        return rfail(FW_ERR_INVALID, "fw_dwconv3x3_nhwc: bad argument");
    return rguard([&] {
        const int co = mode == 1 ? channels / 2 : channels;
        const int strips = (H + fw::DW_ROWS - 1) / fw::DW_ROWS;
        static const bool wide_on = [] {   // FW_REST_DW_WIDE=0: one kernel for all widths (A/B)
            const char* e = getenv("FW_REST_DW_WIDE");
            return !e || atoi(e) != 0;
        }();
        if (wide_on && co >= 512) {
            const int ngr = (co / 8 + 31) / 32;
            int ncb = (W + 7) / 8;
            const int cap = 512 / ngr > 0 ? 512 / ngr : 1;               // about one round of resident blocks over (columns x strips)
            int by = strips;
            if (ncb * by > cap) {
                by = cap / ncb > 0 ? cap / ncb : 1;
                if (ncb * by > cap) ncb = cap;
            }
            const dim3 grid((unsigned)(ncb * ngr), (unsigned)by);
            hipStream_t st = (hipStream_t)stream;
#define FW_DWW(T, MO) hipLaunchKernelGGL((dwconv3x3_nhwc_wide_kernel<T, MO>), grid, dim3(256), 0, st, (const T*)x, ldx, H, W, channels, weight, (T*)out, ldo)
            if (dtype == FW_DTYPE_BF16) { if (mode) FW_DWW(__bf16, 1); else FW_DWW(__bf16, 0); }
            else { if (mode) FW_DWW(_Float16, 1); else FW_DWW(_Float16, 0); }
#undef FW_DWW
            FW_HIP_CHECK(hipGetLastError());
            return;
        }
        const int bx = blocks_for((long)W * (co / 8), 1024);
        const int by_cap = 512 / bx > 0 ? 512 / bx : 1;   // one round of the 512 resident blocks: every block stages the filters in LDS first, so few fat blocks
        const dim3 blocks(bx, strips < by_cap ? strips : by_cap);
        const size_t smem = (size_t)9 * channels * sizeof(float);
        static const bool attr = [] {
            const int cap = 9 * 4096 * (int)sizeof(float);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_nhwc_kernel<__bf16, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_nhwc_kernel<__bf16, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_nhwc_kernel<_Float16, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_nhwc_kernel<_Float16, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            return true;
        }();
        (void)attr;
        hipStream_t st = (hipStream_t)stream;
#define FW_DW(T, MO) hipLaunchKernelGGL((dwconv3x3_nhwc_kernel<T, MO>), blocks, dim3(256), smem, st, (const T*)x, ldx, H, W, channels, weight, (T*)out, ldo)
        if (dtype == FW_DTYPE_BF16) { if (mode) FW_DW(__bf16, 1); else FW_DW(__bf16, 0); }
        else { if (mode) FW_DW(_Float16, 1); else FW_DW(_Float16, 0); }
#undef FW_DW
        FW_HIP_CHECK(hipGetLastError());
    });
}

size_t fw_attn_workspace_floats(int heads, int ch) {
    const size_t dim = (size_t)heads * ch;
    return (size_t)fw::GRAM_MAX_BLOCKS * (dim * ch + 2 * dim);
}

int fw_attn_matrix(int dtype, const void* qkv, long ld, long M, int k_off, int heads, int ch, const float* temperature,
                   float* workspace, float* attn, void* stream) {
    const int dim = heads * ch;
    if (bad_dtype(dtype) || !qkv || !temperature || !workspace || !attn || M < 1 || heads < 1 || (ch != 48 && ch != 96) ||
        dim > 512 || heads * (ch / 3) * (ch / 3) > 8 * 256 || (ld & 7) || (k_off & 7) || ((size_t)qkv & 15))
        return rfail(FW_ERR_INVALID, "fw_attn_matrix: bad argument (16-byte aligned rows expected)");
    return rguard([&] {
        hipStream_t st = (hipStream_t)stream;
        const long chunks = (M + GRAM_PX - 1) / GRAM_PX;
        const int nb = (int)(chunks < GRAM_MAX_BLOCKS ? chunks : GRAM_MAX_BLOCKS);
        const size_t smem = (size_t)2 * GRAM_PX * dim * sizeof(float);
        if (dtype == FW_DTYPE_BF16)
            hipLaunchKernelGGL((attn_gram_kernel<__bf16>), dim3(nb), dim3(256), smem, st, (const __bf16*)qkv, ld, M, k_off, heads, ch, workspace);
        else
            hipLaunchKernelGGL((attn_gram_kernel<_Float16>), dim3(nb), dim3(256), smem, st, (const _Float16*)qkv, ld, M, k_off, heads, ch, workspace);
        const long stride = (long)dim * ch + 2 * dim;
        const int G = nb < 32 ? 1 : 32;
        if (G > 1)
            hipLaunchKernelGGL(attn_reduce_kernel, dim3((unsigned)((stride + 255) / 256), G), dim3(256), 0, st, workspace, nb, stride, G);
        hipLaunchKernelGGL(attn_finish_kernel, dim3(heads * ch), dim3(128), 0, st, (const float*)workspace, G > 1 ? G : nb, stride, heads, ch, temperature, attn);
        FW_HIP_CHECK(hipGetLastError());
    });
}

size_t fw_attn_qk_scratch_elems(long pixels, int heads, int ch) {
    const long mp = (pixels + 31) / 32 * 32;
    return (size_t)2 * mp * heads * ch;
}

int fw_attn_matrix_mfma(int dtype, const void* qkv, long ld, long M, int k_off, int heads, int ch, const float* temperature,
                        float* workspace, void* qk_scratch, float* attn, void* stream) {
    const int dim = heads * ch;
    if (bad_dtype(dtype) || !qkv || !temperature || !workspace || !qk_scratch || !attn || M < 1 || heads < 1 || (ch != 48 && ch != 96) ||
        dim > 512 || (ld & 7) || (k_off & 7) || ((size_t)qkv & 15) || ((size_t)qk_scratch & 15))
        return rfail(FW_ERR_INVALID, "fw_attn_matrix_mfma: bad argument");
    return rguard([&] {
        hipStream_t st = (hipStream_t)stream;
        const long Mp = (M + 31) / 32 * 32;
        const long steps = Mp / 32;
        long nbl = steps / 8;
        const int nb = (int)(nbl < 1 ? 1 : (nbl > GRAM_MAX_BLOCKS ? GRAM_MAX_BLOCKS : nbl));
        const int tb = blocks_for((Mp / 8) * (dim / 8) * 2, 2048);
        const size_t half = (size_t)Mp * dim;
#define FW_GM(T)                                                                                                              \
    do {                                                                                                                      \
        T* qT = (T*)qk_scratch;                                                                                               \
        T* kT = qT + half;                                                                                                    \
        hipLaunchKernelGGL((qk_transpose_kernel<T>), dim3(tb), dim3(256), 0, st, (const T*)qkv, ld, M, Mp, k_off, dim, qT, kT); \
        if (ch == 48)                                                                                                         \
            hipLaunchKernelGGL((attn_gram_mfma_kernel<T, 3>), dim3(nb, heads), dim3(256), 0, st, (const T*)qT, (const T*)kT, Mp, heads, dim, workspace); \
        else                                                                                                                  \
            hipLaunchKernelGGL((attn_gram_mfma_kernel<T, 6>), dim3(nb, heads), dim3(256), 0, st, (const T*)qT, (const T*)kT, Mp, heads, dim, workspace); \
    } while (0)
        if (dtype == FW_DTYPE_BF16) FW_GM(__bf16); else FW_GM(_Float16);
#undef FW_GM
        const long stride = (long)dim * ch + 2 * dim;
        const int G = nb < 32 ? 1 : 32;
        if (G > 1)
            hipLaunchKernelGGL(attn_reduce_kernel, dim3((unsigned)((stride + 255) / 256), G), dim3(256), 0, st, workspace, nb, stride, G);
        hipLaunchKernelGGL(attn_finish_kernel, dim3(heads * ch), dim3(128), 0, st, (const float*)workspace, G > 1 ? G : nb, stride, heads, ch, temperature, attn);
        FW_HIP_CHECK(hipGetLastError());
    });
}

// The attention matrices from q and k already held as [pixel group of 8][ldc channels][8 pixels] (pw_dw_fused.hip writes them that
// way, in its own pixel order and zero padded per tile - the Gram sum does not care about the order): Gram partials, reduction,
// softmax.  Mp: pixels including padding, a multiple of 32.
}  // extern "C"
namespace fw {
void launch_attn_matrix_from_transposed(DType dt, const void* qT, const void* kT, long Mp, int ldc, int heads, int ch, const float* temperature,
                                        float* workspace, float* attn, hipStream_t st) {
    const int dim = heads * ch;
    if (!qT || !kT || !temperature || !workspace || !attn || Mp < 32 || (Mp & 31) || heads < 1 || (ch != 48 && ch != 96) || dim > 512 || ldc < dim)
        throw Error(FW_ERR_INVALID, "attn_matrix_from_transposed: bad argument");
    const long steps = Mp / 32;
    const long nbl = steps / 8;
    const int nb = (int)(nbl < 1 ? 1 : (nbl > GRAM_MAX_BLOCKS ? GRAM_MAX_BLOCKS : nbl));
#define FW_GM(T)                                                                                                                        \
    do {                                                                                                                                \
        if (ch == 48)                                                                                                                   \
            hipLaunchKernelGGL((attn_gram_mfma_kernel<T, 3>), dim3(nb, heads), dim3(256), 0, st, (const T*)qT, (const T*)kT, Mp, heads, ldc, workspace); \
        else                                                                                                                            \
            hipLaunchKernelGGL((attn_gram_mfma_kernel<T, 6>), dim3(nb, heads), dim3(256), 0, st, (const T*)qT, (const T*)kT, Mp, heads, ldc, workspace); \
    } while (0)
    if (dt == DT_BF16) FW_GM(__bf16); else FW_GM(_Float16);
#undef FW_GM
    const long stride = (long)dim * ch + 2 * dim;
    const int G = nb < 32 ? 1 : 32;
    if (G > 1) hipLaunchKernelGGL(attn_reduce_kernel, dim3((unsigned)((stride + 255) / 256), G), dim3(256), 0, st, workspace, nb, stride, G);
    hipLaunchKernelGGL(attn_finish_kernel, dim3(heads * ch), dim3(128), 0, st, (const float*)workspace, G > 1 ? G : nb, stride, heads, ch, temperature, attn);
    FW_HIP_CHECK(hipGetLastError());
}
}  // namespace fw
extern "C" {

int fw_attn_pack(int dtype, const float* attn, int heads, int ch, int k_pad, void* packed, void* stream) {
    if (bad_dtype(dtype) || !attn || !packed || heads < 1 || ch < 8 || k_pad < heads * ch || (k_pad & 31))
        return rfail(FW_ERR_INVALID, "fw_attn_pack: bad argument");
    return rguard([&] {
        const long total = (long)(k_pad / 32) * 2 * (k_pad / 32) * 64 * 8;
        const int blocks = blocks_for(total, 1024);
        if (dtype == FW_DTYPE_BF16)
            hipLaunchKernelGGL((attn_pack_kernel<__bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, attn, heads, ch, k_pad, (__bf16*)packed);
        else
            hipLaunchKernelGGL((attn_pack_kernel<_Float16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, attn, heads, ch, k_pad, (_Float16*)packed);
        FW_HIP_CHECK(hipGetLastError());
    });
}

int fw_attn_proj_pack(int dtype, const float* attn, const float* proj_weight, int heads, int ch, int k_pad, int cout_tiles, void* packed,
                      void* stream) {
    if (bad_dtype(dtype) || !attn || !proj_weight || !packed || heads < 1 || ch < 8 || k_pad < heads * ch || (k_pad & 31) || cout_tiles < 1 ||
        32 * cout_tiles < heads * ch)
        return rfail(FW_ERR_INVALID, "fw_attn_proj_pack: bad argument");
    return rguard([&] {
        const long total = (long)(k_pad / 32) * 2 * cout_tiles * 64 * 8;
        const int blocks = blocks_for(total, 1024);
        if (dtype == FW_DTYPE_BF16)
            hipLaunchKernelGGL((attn_proj_pack_kernel<__bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, attn, proj_weight, heads, ch, k_pad,
                               cout_tiles, (__bf16*)packed);
        else
            hipLaunchKernelGGL((attn_proj_pack_kernel<_Float16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, attn, proj_weight, heads, ch, k_pad,
                               cout_tiles, (_Float16*)packed);
        FW_HIP_CHECK(hipGetLastError());
    });
}

int fw_attn_apply(int dtype, const void* qkv, long ld, long M, int v_off, int heads, int ch, const float* attn, void* out,
                  long ldo, int zero_to, void* stream) {
    const int dim = heads * ch;
    if (bad_dtype(dtype) || !qkv || !attn || !out || M < 1 || (ch & 7) || (zero_to & 7) || zero_to < dim || zero_to > ldo)
        return rfail(FW_ERR_INVALID, "fw_attn_apply: bad argument");
    return rguard([&] {
        const size_t smem = (size_t)dim * ch * sizeof(float);
        static const bool attr = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_apply_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_apply_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            return true;
        }();
        (void)attr;
        const int blocks = blocks_for(M * (zero_to / 8), 2048);
        hipStream_t st = (hipStream_t)stream;
        if (dtype == FW_DTYPE_BF16)
            hipLaunchKernelGGL((attn_apply_kernel<__bf16>), dim3(blocks), dim3(256), smem, st, (const __bf16*)qkv, ld, M, v_off, heads, ch, attn, (__bf16*)out, ldo, zero_to);
        else
            hipLaunchKernelGGL((attn_apply_kernel<_Float16>), dim3(blocks), dim3(256), smem, st, (const _Float16*)qkv, ld, M, v_off, heads, ch, attn, (_Float16*)out, ldo, zero_to);
        FW_HIP_CHECK(hipGetLastError());
    });
}

int fw_attn_softmax_rows(int dtype, const void* q, long q_stride, const void* k, long k_stride, long pixels, int d, void* p,
                         long p_stride, void* stream) {
    if (bad_dtype(dtype) || !q || !k || !p || pixels < 1 || d < 1 || d > 8 || (q_stride & 7) || (k_stride & 7) || p_stride < pixels ||
        ((size_t)q & 15) || ((size_t)k & 15))
        return rfail(FW_ERR_INVALID, "fw_attn_softmax_rows: bad argument");
    return rguard([&] {
        const int blocks = blocks_for(pixels * 64, 2048);
        if (dtype == FW_DTYPE_BF16)
            hipLaunchKernelGGL((attn_softmax_rows_kernel<__bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)q, q_stride,
                               (const __bf16*)k, k_stride, pixels, d, (__bf16*)p, p_stride);
        else
            hipLaunchKernelGGL((attn_softmax_rows_kernel<_Float16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)q,
                               q_stride, (const _Float16*)k, k_stride, pixels, d, (_Float16*)p, p_stride);
        FW_HIP_CHECK(hipGetLastError());
    });
}

int fw_pack_pointwise_transposed(int dtype, const void* src, long src_stride, long k_valid, int cout, int k_pad, void* packed,
                                 void* stream) {
    if (bad_dtype(dtype) || !src || !packed || k_valid < 1 || cout < 1 || k_pad < k_valid || (k_pad & 31))
        return rfail(FW_ERR_INVALID, "fw_pack_pointwise_transposed: bad argument");
    return rguard([&] {
        const long total = (long)(k_pad / 32) * 2 * ((cout + 31) / 32) * 64;
        const int blocks = blocks_for(total, 2048);
        if (dtype == FW_DTYPE_BF16)
            hipLaunchKernelGGL((pack_pointwise_t_kernel<__bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)src,
                               src_stride, k_valid, cout, k_pad, (__bf16*)packed);
        else
            hipLaunchKernelGGL((pack_pointwise_t_kernel<_Float16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src,
                               src_stride, k_valid, cout, k_pad, (_Float16*)packed);
        FW_HIP_CHECK(hipGetLastError());
    });
}

int fw_pixel_shuffle2_f32(const float* src, long src_stride, int low_h, int low_w, int channels, float* dst, long dst_stride,
                          int dst_coff, int unshuffle, void* stream) {
    if (!src || !dst || low_h < 1 || low_w < 1 || channels < 1) return rfail(FW_ERR_INVALID, "fw_pixel_shuffle2_f32: bad argument");
    return rguard([&] {
        const int blocks = blocks_for((long)low_h * low_w * channels * 4, 4096);
        hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, src_stride, low_h, low_w,
                           channels, dst, dst_stride, dst_coff, unshuffle);
        FW_HIP_CHECK(hipGetLastError());
    });
}

int fw_copy_channels_f32(const float* src, long src_stride, long M, int channels, float* dst, long dst_stride, int dst_coff,
                         void* stream) {
    if (!src || !dst || M < 1 || channels < 1) return rfail(FW_ERR_INVALID, "fw_copy_channels_f32: bad argument");
    return rguard([&] {
        hipLaunchKernelGGL(copy_channels_kernel, dim3(blocks_for(M * channels, 4096)), dim3(256), 0, (hipStream_t)stream, src,
                           src_stride, M, channels, dst, dst_stride, dst_coff);
        FW_HIP_CHECK(hipGetLastError());
    });
}

int fw_f32_to_planar(int dtype, const float* x, long M, int channels, void* out, void* stream) {
    if (bad_dtype(dtype) || !x || !out || M < 1 || channels < 32 || (channels & 31))
        return rfail(FW_ERR_INVALID, "fw_f32_to_planar: bad argument");
    return rguard([&] { launch_f32_to_planar((DType)dtype, x, M, channels, out, (hipStream_t)stream); });
}

int fw_tap_post_u8(const uint8_t* in_bgr, const float* rgb, int H, int W, int padded_w, int rgb_cstride, uint8_t* out_bgr,
                   float* out_rgb_f32, void* stream) {
    if (!in_bgr || !rgb || (!out_bgr && !out_rgb_f32) || H < 1 || W < 1 || rgb_cstride < 3)
        return rfail(FW_ERR_INVALID, "fw_tap_post_u8: bad argument");
    return rguard([&] { launch_tap_post(in_bgr, rgb, H, W, padded_w, rgb_cstride, out_bgr, out_rgb_f32, (hipStream_t)stream); });
}

int fw_flow_accumulate_u8(const uint8_t* frame_bgr, const float* flow_x, const float* flow_y, const float* weight_map,
                          double weight_scale, const float* magnitude, float motion_threshold, int inverse, int H, int W,
                          double* accumulated, double* weight_sum, void* stream) {
    if (!frame_bgr || !accumulated || !weight_sum || H < 1 || W < 1 || (!flow_x != !flow_y))
        return rfail(FW_ERR_INVALID, "fw_flow_accumulate_u8: bad argument");
    return rguard([&] {
        launch_flow_accumulate(frame_bgr, flow_x, flow_y, weight_map, weight_scale, magnitude, motion_threshold, inverse, H, W,
                               accumulated, weight_sum, (hipStream_t)stream);
    });
}

int fw_flow_accumulate_finish_u8(const double* accumulated, const double* weight_sum, int H, int W, uint8_t* out_bgr,
                                 void* stream) {
    if (!accumulated || !weight_sum || !out_bgr || H < 1 || W < 1) return rfail(FW_ERR_INVALID, "fw_flow_accumulate_finish_u8: bad argument");
    return rguard([&] { launch_flow_accumulate_finish(accumulated, weight_sum, (long)H * W, out_bgr, (hipStream_t)stream); });
}

}  // extern "C"
