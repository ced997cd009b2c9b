// NAFNet building blocks (kernel set K6 of SURVEY.md §8a) for the TAP temporal-denoise path.
//
// The reference runs these through third-party `basicsr.archs.nafnet_arch.NAFNet` (call site reference
// src/framewright/processors/tap_denoise.py:335-364, forward at :458); the architecture restated here is the one
// recorded in SURVEY.md §A.3 (NAFBlock: LayerNorm2d -> 1x1 -> depthwise 3x3 -> SimpleGate -> SCA -> 1x1 -> beta
// residual -> LayerNorm2d -> 1x1 -> SimpleGate -> 1x1 -> gamma residual; 2x2/s2 down, 1x1 + PixelShuffle up).
//
// Data layout: the residual stream is fp32 NHWC ([pixel][C]); everything that feeds an MFMA is operand-typed NHWC.
//   * pointwise_mfma_kernel : 1x1 conv / 2x2-s2 conv / 1x1+PixelShuffle as one MFMA GEMM D[cout][pixel], pixel on the
//                             lane (same orientation as conv3x3_mfma.hip), fused SimpleGate / residual epilogues.
//   * layernorm2d_kernel    : per-pixel LayerNorm over channels (HBM-bound, 16 / 32 / 64 lanes per pixel, shuffles).
//   * dwconv3x3_gate_kernel : depthwise 3x3 + SimpleGate + per-channel partial sums for SCA (HBM-bound).
//   * sca_kernel            : global-average-pool finish + 1x1 conv (a CxC mat-vec).
#include "fw_internal.h"

namespace fw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Tr;
template <>
struct Tr<__bf16> {
    using v8 = bf16x8;
    using v4 = bf16x4;
    static __device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                       0, 0, 0);
    }
};
template <>
struct Tr<_Float16> {
    using v8 = f16x8;
    using v4 = f16x4;
    static __device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    }
};

template <typename T>
__device__ __forceinline__ uint4 pack8f(const float* v) {
    typename Tr<T>::v8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (T)v[i];
    return __builtin_bit_cast(uint4, o);
}
template <typename T>
__device__ __forceinline__ void unpack8f(uint4 u, float* v) {
    typename Tr<T>::v8 o = __builtin_bit_cast(typename Tr<T>::v8, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)o[i];
}
template <typename T>
__device__ __forceinline__ uint2 pack4f(float a, float b, float c, float d) {
    typename Tr<T>::v4 v = {(T)a, (T)b, (T)c, (T)d};
    return __builtin_bit_cast(uint2, v);
}

// =====================================================================================================
// Pointwise MFMA GEMM:  D[n][m] = sum_k W[n][k] * A[m][k]
//   block = 256 threads = 4 waves; tile = 256 pixels x 32*CT couts; wave w owns pixels [64w, 64w+64) (2 MFMA
//   pixel tiles) and all CT cout tiles.  K walked in 32-channel chunks: the A tile (256 px x 64 B) is staged
//   global -> registers -> LDS (fp32 -> operand conversion and the optional per-channel SCA scale happen in that
//   pass), the chunk's weight fragments (2*CT KiB, packed like the conv weights with one "tap") go through LDS too.
// =====================================================================================================
constexpr int PW_PX = 256;

template <typename T, int CT, int MODE>
__global__ __launch_bounds__(256, 2) void pointwise_mfma_kernel(const PointwiseParams p) {
    // two staging buffers (round 3: one barrier per chunk - chunk c + 1 is written while other waves still multiply chunk c; with one buffer
    // and two barriers a chunk cost 2.9 us where a CU holds one workgroup, 64 chunks of the deepest 2x2 stride-2 conv took 189 us)
    constexpr int PW_STAGE = PW_PX * 4 + 2 * CT * 64;
    __shared__ __attribute__((aligned(16))) uint4 lds_all[2 * PW_STAGE];   // one array: the turned epilogues reuse it
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int n_tiles = (p.N_tiles + CT - 1) / CT;  // blocks along couts
    const int bm = blockIdx.x / n_tiles;
    const int bn = blockIdx.x - bm * n_tiles;
    const long m0 = (long)bm * PW_PX;

    f32x16 acc[2][CT];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[pt][ct][i] = 0.f;

    // staging plan: thread -> (pixel, 16-byte slot), 4 pieces per thread
    const int chunks = p.K / 32;
    // LayerNorm2d fused into the staging (K = 64): the four threads that stage a pixel hold its 64 fp32 channels between them
    // (2 chunks x 8 values each), reduce mean and variance over their quad with two shuffles, and normalise on the way to
    // LDS - the LayerNorm kernel's pass over HBM (read 256 B, write 128 B per pixel) and the typed re-read are gone.
    const bool ln = p.ln_w != nullptr;   // uniform
    float lnv[4][2][8];
    float ln_mean[4], ln_rstd[4];
    if (ln) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = (tid + 256 * i) >> 2, sq = tid & 3;
            const long m = m0 + px;
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
                if (m < p.M) {
                    const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.a) + m * p.lda + c * 32 + sq * 8);
                    lo = src[0];
                    hi = src[1];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lnv[i][c][j] = lo[j];
                    lnv[i][c][4 + j] = hi[j];
                    sum += lo[j] + hi[j];
                }
            }
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            ln_mean[i] = sum * (1.f / 64.f);
            float qq = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = lnv[i][c][j] - ln_mean[i];
                    qq += d * d;
                }
            qq += __shfl_xor(qq, 1);
            qq += __shfl_xor(qq, 2);
            ln_rstd[i] = 1.0f / sqrtf(qq * (1.f / 64.f) + p.ln_eps);
        }
    }
    constexpr int NWV = (2 * CT * 64 + 255) / 256;
    auto wload = [&](int c, uint4 (&wv)[NWV]) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 2 * CT * 64) {
                // fragment order [chunk][ks][cout tile][lane]; this block's cout tiles start at bn*CT (MODE gate: see below)
                const int ks = idx / (CT * 64), rem = idx - ks * (CT * 64);
                const int ct = rem >> 6, ln_ = rem & 63;
                int tile = bn * CT + ct;
                if (MODE == PW_GATE) tile = bn * (CT / 2) + (ct >> 1) + (ct & 1) * (p.N_tiles / 2);  // pair n with n + N/2
                wv[i] = tile < p.N_tiles
                            ? reinterpret_cast<const uint4*>(p.wpk)[(((size_t)c * 2 + ks) * p.N_tiles + tile) * 64 + ln_]
                            : make_uint4(0, 0, 0, 0);
            }
        }
    };
    int nstage = 0;   // chunks staged so far: chunk n lives in buffer n & 1
    auto stage = [&](const uint4 (&v)[4], const uint4 (&wv)[NWV]) {
        uint4* const lds_a = lds_all + (nstage & 1) * PW_STAGE;
        uint4* const lds_w = lds_a + PW_PX * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int px = idx >> 2, s = idx & 3;
            lds_a[px * 4 + (s ^ ((px >> 2) & 3))] = v[i];
        }
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 2 * CT * 64) lds_w[idx] = wv[i];
        }
        // every wave has written chunk n and is done with the MFMAs of chunk n - 1; the writes of chunk n + 1 (other buffer) may start while
        // slower waves multiply chunk n, those of chunk n + 2 (this buffer) come after the next barrier
        __syncthreads();
        ++nstage;
    };
    auto mma = [&]() {
        const uint4* const lds_a = lds_all + ((nstage - 1) & 1) * PW_STAGE;
        const uint4* const lds_w = lds_a + PW_PX * 4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 wf[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wf[ct] = lds_w[(ks * CT + ct) * 64 + lane];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int px = wave * 64 + pt * 32 + r;
                const uint4 xf = lds_a[px * 4 + ((2 * ks + h) ^ ((px >> 2) & 3))];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[pt][ct] = Tr<T>::mfma(wf[ct], xf, acc[pt][ct]);
            }
        }
    };
    if (ln) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = c * 32 + (tid & 3) * 8;
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (lnv[i][c][j] - ln_mean[i]) * ln_rstd[i] * p.ln_w[kk + j] + p.ln_b[kk + j];
                v[i] = (m0 + ((tid + 256 * i) >> 2) < p.M) ? pack8f<T>(f) : make_uint4(0, 0, 0, 0);
            }
            uint4 wv[NWV];
            wload(c, wv);
            stage(v, wv);
            mma();
        }
    }
    // The other inputs: chunk c + 1's loads (operand rows and weight fragments) are issued right after chunk c is staged, so they are
    // in flight while its MFMAs run (they used to be issued after them: load latency + MFMAs per chunk, exposed whenever a CU holds
    // one workgroup - the 16 k- and 4 k-pixel levels of Restormer, NAFNet's deep levels).
    // A loads in two steps (round 3): `aissue` only loads - raw bits, rows clamped to M - 1 so that no branch separates the loads - and
    // `aconv` turns them into operand fragments (fp32 -> operand type, SCA scale, zeros behind M) when the chunk is staged.  With the
    // conversion inside the load step every piece waited for its own loads: four exposed latencies per chunk (the deepest 2x2 stride-2
    // conv of NAFNet, 64 chunks on one workgroup per CU: 2.9 us per chunk, 190 us), and the "prefetch" of the next chunk hid nothing.
    // The per-pixel part of the addresses (64-bit divisions of the 2x2 gather) is computed once, not per chunk.
    long arow[4];        // element offset of the piece's pixel row (2x2 gather: of the pixel's top-left source pixel)
    bool aok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int px = (tid + 256 * i) >> 2;
        long m = m0 + px;
        aok[i] = m < p.M;
        if (!aok[i]) m = p.M - 1;
        if (p.gather2x2) {
            const int Wo = p.Win >> 1;
            const int y = (int)(m / Wo), x = (int)(m - (long)y * Wo);
            arow[i] = ((long)(2 * y) * p.Win + 2 * x) * p.lda;
        } else {
            arow[i] = m * p.lda;
        }
    }
    auto aoff = [&](int c, int i, int* kk_out) -> long {   // element offset of piece i of chunk c; *kk_out = its first channel (for a_scale)
        int kk = c * 32 + (tid & 3) * 8;
        long off = arow[i];
        if (p.gather2x2) {
            // 2x2 stride-2 conv: k = (dy*2+dx)*Cin + ci ; output pixel m = (y, x) of the half-size grid
            const int sub = kk / p.Cin;
            kk -= sub * p.Cin;
            off += ((long)(sub >> 1) * p.Win + (sub & 1)) * p.lda;
            *kk_out = c * 32 + (tid & 3) * 8;
            return off + kk;
        }
        *kk_out = kk;
        return off + kk;
    };
    auto aissue = [&](int c, uint4 (&q)[4][2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int kk;
            const long off = aoff(c, i, &kk);
            if (p.a_f32) {
                const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(p.a) + off);
                q[i][0] = src[0];
                q[i][1] = src[1];
            } else {
                q[i][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.a) + off);
            }
        }
    };
    auto aconv = [&](int c, const uint4 (&q)[4][2], uint4 (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = c * 32 + (tid & 3) * 8;
            float f[8];
            if (p.a_f32) {
                const f32x4 lo = __builtin_bit_cast(f32x4, q[i][0]), hi = __builtin_bit_cast(f32x4, q[i][1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f[j] = lo[j];
                    f[4 + j] = hi[j];
                }
                if (p.a_scale) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] *= p.a_scale[kk + j];
                }
                v[i] = pack8f<T>(f);
            } else {
                v[i] = q[i][0];
                if (p.a_scale) {
                    unpack8f<T>(v[i], f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] *= p.a_scale[kk + j];
                    v[i] = pack8f<T>(f);
                }
            }
            if (!aok[i]) v[i] = make_uint4(0, 0, 0, 0);
        }
    };
    if (!ln) {
        uint4 q[4][2], v[4], wv[NWV];
        aissue(0, q);
        wload(0, wv);
        for (int c = 0; c < chunks; ++c) {
            aconv(c, q, v);
            stage(v, wv);
            if (c + 1 < chunks) {
                aissue(c + 1, q);
                wload(c + 1, wv);
            }
            mma();
        }
    }

    // ---- epilogue: lane holds pixel (wave*64 + pt*32 + r), couts 32*tile + 8g + 4h + j ------------------------------
#ifndef FW_PW_RES_DIRECT
    // (round 3: the fp32 store of the 2x2 stride-2 convs and the PixelShuffle + skip epilogue of the up convs take the same way out - straight
    //  from the accumulators NAFNet's four down and four up convs, HBM-bound by construction, moved 2.1 / 2.7 TB/s)
    constexpr bool TURNED = MODE == PW_RESIDUAL || MODE == PW_SHUFFLE_UP || MODE == PW_STORE;
    if constexpr (TURNED) if (MODE != PW_STORE || (p.out_f32 && !p.out_typed)) {
        // The residual epilogue moves four times the bytes of the K loop at K = N (fp32 stream in and out against a typed operand),
        // and straight from the accumulator layout a wave-instruction touched 32 pixel rows with 32 bytes each.  Here each 32-pixel
        // x 32-channel accumulator tile is turned round in a wave-private LDS slice ([32][36] floats in the staging buffers the K
        // loop is done with; conflict-free 16-byte writes), after which 8 lanes own one pixel's 128 contiguous bytes: the stream is
        // read and written in whole cache lines.  The reads of a 32-pixel group are issued before its tiles are turned.
        __syncthreads();                                            // every wave is done with lds_a / lds_w
        static_assert(2 * PW_STAGE * 16 >= 4 * 32 * 36 * 4, "staging buffers too small for the epilogue slices");
        float* slice = reinterpret_cast<float*>(lds_all) + wave * (32 * 36);
        const int piece = lane & 7, prow = lane >> 3;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const long mg = m0 + wave * 64 + pt * 32;                 // first pixel of the group (wave-uniform)
            if (mg >= p.M) continue;
            // where this lane's four pixels (rows prow + 8 i of the group) and its four channels of a tile live in the fp32 stream
            long orow[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long m = mg + prow + 8 * i;
                if constexpr (MODE == PW_SHUFFLE_UP) {
                    const int Wl = p.Win;
                    const long mm = m < p.M ? m : p.M - 1;
                    const int y = (int)(mm / Wl), x = (int)(mm - (long)y * Wl);
                    orow[i] = ((long)(2 * y) * (2 * Wl) + 2 * x) * p.ldf;      // sub-position (dy, dx) added per tile
                } else {
                    orow[i] = m * p.ldf;
                }
            }
            auto ocol = [&](int tile) -> long {   // column offset of channel 32 tile + 4 piece (PW_SHUFFLE_UP: n = sub * Cup + co, Cup % 32 == 0)
                const int n = 32 * tile + 4 * piece;
                if constexpr (MODE == PW_SHUFFLE_UP) {
                    const int Cup = p.N_tiles * 8;
                    const int sub = n / Cup, co = n - sub * Cup;
                    return ((long)(sub >> 1) * (2 * p.Win) + (sub & 1)) * p.ldf + co;
                } else {
                    return n;
                }
            };
            f32x4 rs[CT][4];
            if constexpr (MODE != PW_STORE) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int tile = bn * CT + ct;
                    if (tile >= p.N_tiles) continue;
                    const long oc = ocol(tile);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const long m = mg + prow + 8 * i;
                        rs[ct][i] = m < p.M ? *reinterpret_cast<const f32x4*>(p.res_f32 + orow[i] + oc) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int tile = bn * CT + ct;
                if (tile >= p.N_tiles) continue;
                const int n = 32 * tile + 4 * piece;
                const long oc = ocol(tile);
                const f32x4 bs = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 sc = {1.f, 1.f, 1.f, 1.f};
                if constexpr (MODE == PW_RESIDUAL) sc = *reinterpret_cast<const f32x4*>(p.chan_scale + n);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[pt][ct][4 * g], acc[pt][ct][4 * g + 1], acc[pt][ct][4 * g + 2], acc[pt][ct][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(slice + r * 36 + 8 * g + 4 * h) = v;
                }
                // (a wave reads back only what it wrote: no workgroup barrier; the compiler orders the LDS accesses of one wave)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const long m = mg + prow + 8 * i;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(slice + (prow + 8 * i) * 36 + 4 * piece);
                    f32x4 of;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float o = a[j] + bs[j];
                        if constexpr (MODE == PW_RESIDUAL) of[j] = rs[ct][i][j] + o * sc[j];
                        else if constexpr (MODE == PW_SHUFFLE_UP) of[j] = o + rs[ct][i][j];
                        else of[j] = o;
                    }
                    if (m < p.M) *reinterpret_cast<f32x4*>(p.out_f32 + orow[i] + oc) = of;
                }
            }
        }
        return;
    }
#endif
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const long m = m0 + wave * 64 + pt * 32 + r;
        if (m >= p.M) continue;
        if constexpr (MODE == PW_GATE) {
            // SimpleGate fused: tiles (2q, 2q+1) hold channels n and n + N/2 of the same pixel -> out[n] = x1 * x2
#pragma unroll
            for (int q = 0; q < CT / 2; ++q) {
                const int n_base = 32 * (bn * (CT / 2) + q);
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {  // 16-byte stores: see PW_STORE below for the lane exchange
                    uint2 pk[2];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int g = 2 * gp + gg;
                        const int n = n_base + 8 * g + 4 * h;
                        float o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            o[j] = (acc[pt][2 * q][4 * g + j] + p.bias[n + j]) *
                                   (acc[pt][2 * q + 1][4 * g + j] + p.bias[n + j + p.N_tiles * 16]);
                        pk[gg] = pack4f<T>(o[0], o[1], o[2], o[3]);
                    }
                    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                    const u32x2_t sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                    const u32x2_t sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                    *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out_typed) + m * p.ldo + n_base + 16 * gp + 8 * h) =
                        make_uint4(sx[0], sy[0], sx[1], sy[1]);
                }
            }
        } else {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int tile = bn * CT + ct;
                if (tile >= p.N_tiles) continue;
                if constexpr (MODE == PW_STORE) {
                    // typed output, 16 bytes per lane: lanes l and l ^ 32 hold channels 8g + 4h.. of the same pixel; one
                    // v_permlane32_swap per dword pairs the fragments of groups g, g + 1 so that lane h = 0 owns channels
                    // 8g..8g+7 and lane h = 1 channels 8g+8..8g+15 (it was four scattered 8-byte stores per 32 channels)
                    if (p.out_typed && !p.out_f32) {
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            uint2 pk[2];
#pragma unroll
                            for (int gg = 0; gg < 2; ++gg) {
                                const int g = 2 * gp + gg;
                                const int n = 32 * tile + 8 * g + 4 * h;
                                float o[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) o[j] = acc[pt][ct][4 * g + j] + (p.bias ? p.bias[n + j] : 0.f);
                                pk[gg] = pack4f<T>(o[0], o[1], o[2], o[3]);
                            }
                            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                            const u32x2_t sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                            const u32x2_t sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                            *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out_typed) + m * p.ldo + 32 * tile + 16 * gp + 8 * h) =
                                make_uint4(sx[0], sy[0], sx[1], sy[1]);
                        }
                        continue;
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 32 * tile + 8 * g + 4 * h;
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = acc[pt][ct][4 * g + j] + (p.bias ? p.bias[n + j] : 0.f);
                    if constexpr (MODE == PW_STORE) {
                        if (p.out_typed)
                            *reinterpret_cast<uint2*>(reinterpret_cast<T*>(p.out_typed) + m * p.ldo + n) =
                                pack4f<T>(o[0], o[1], o[2], o[3]);
                        if (p.out_f32) {
                            f32x4 of = {o[0], o[1], o[2], o[3]};
                            *reinterpret_cast<f32x4*>(p.out_f32 + m * p.ldf + n) = of;
                        }
                    } else if constexpr (MODE == PW_RESIDUAL) {
                        // y = res + x * chan_scale   (beta / gamma residual of the NAFBlock)
                        const f32x4 rs = *reinterpret_cast<const f32x4*>(p.res_f32 + m * p.ldf + n);
                        f32x4 of;
#pragma unroll
                        for (int j = 0; j < 4; ++j) of[j] = rs[j] + o[j] * p.chan_scale[n + j];
                        *reinterpret_cast<f32x4*>(p.out_f32 + m * p.ldf + n) = of;
                    } else {  // PW_SHUFFLE_UP: n = (dy*2+dx)*Cup + co (weights permuted at pack time); m = (y, x) low-res
                        const int Cup = p.N_tiles * 8;  // N / 4
                        const int sub = n / Cup, co = n - sub * Cup;
                        const int Wl = p.Win;
                        const int y = (int)(m / Wl), x = (int)(m - (long)y * Wl);
                        const long mo = (long)(2 * y + (sub >> 1)) * (2 * Wl) + (2 * x + (sub & 1));
                        const f32x4 sk = *reinterpret_cast<const f32x4*>(p.res_f32 + mo * p.ldf + co);
                        f32x4 of = {o[0] + sk[0], o[1] + sk[1], o[2] + sk[2], o[3] + sk[3]};
                        *reinterpret_cast<f32x4*>(p.out_f32 + mo * p.ldf + co) = of;
                    }
                }
            }
        }
    }
}

// =====================================================================================================
// The same GEMM for FEW pixels (Restormer's 192- / 384-channel levels of a 512 x 512 tile: 16 k / 4 k pixels, K up to 1024).
// The kernel above gives such a launch 16 - 64 workgroups for 256 CUs, each walking K in 32-channel chunks with one exposed load
// latency and two barriers per chunk (27 / 55 us at 4096 pixels, K = 384 / 1024).  Here
//   * a workgroup owns 64 pixels x 128 output channels and its four waves split the OUTPUT CHANNELS (wave w: cout tile 4 bn + w,
//     both 32-pixel groups): four times the workgroups, and a wave's weight fragments are its own - they go from global memory
//     straight into registers (1 KiB per wave-instruction, no LDS);
//   * K is walked in stages of 8 chunks (256 channels): a pixel's 512 contiguous bytes are loaded by 32 lanes, staged through LDS
//     in the swizzled image of the kernel above, and the next stage's loads (A and weights: 24 x 16 bytes per lane) are in flight
//     while the 32 MFMAs of this stage run: K = 1024 costs 4 load latencies, not 32;
//   * epilogues: the typed 16-byte store and the turned residual epilogue of the kernel above, per wave.
// Typed operand A only (no fp32 / gather / SCA scale / fused LayerNorm): launch_pointwise routes the rest to the kernel above.
// =====================================================================================================
constexpr int PWS_NS = 8;      // chunks per stage
constexpr int PWS_PX = 64;     // pixels per workgroup

template <typename T, int MODE>
__global__ __launch_bounds__(256) void pointwise_small_kernel(const PointwiseParams p) {
    __shared__ __attribute__((aligned(16))) uint4 lds_a[PWS_NS * PWS_PX * 4];   // [chunk][pixel][4 slots], 32 KiB
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int n_blocks = (p.N_tiles + 3) / 4;
    const int bm = blockIdx.x / n_blocks, bn = blockIdx.x - bm * n_blocks;
    const long m0 = (long)bm * PWS_PX;
    const int tile = bn * 4 + wave;
    const bool live = tile < p.N_tiles;      // wave-uniform: a wave without a tile still stages A and meets the barriers
    const int chunks = p.K / 32, stages = (chunks + PWS_NS - 1) / PWS_NS;

    f32x16 acc[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[pt][i] = 0.f;

    uint4 av[8], wv[PWS_NS][2];
    auto fetch = [&](int s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {        // slot idx = tid + 256 i: pixel idx >> 5, 16-byte slot idx & 31 of the stage's 512 bytes
            const int idx = tid + 256 * i, px = idx >> 5, sl = idx & 31;
            const long m = m0 + px;
            const int kk = (s * PWS_NS) * 32 + sl * 8;
            av[i] = (m < p.M && kk < p.K) ? *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.a) + m * p.lda + kk) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < PWS_NS; ++c)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int cg = s * PWS_NS + c;
                wv[c][ks] = (live && cg < chunks) ? reinterpret_cast<const uint4*>(p.wpk)[(((size_t)cg * 2 + ks) * p.N_tiles + tile) * 64 + lane]
                                                  : make_uint4(0, 0, 0, 0);
            }
    };
    fetch(0);
    for (int s = 0; s < stages; ++s) {
        __syncthreads();                      // the previous stage's fragments are read
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, px = idx >> 5, sl = idx & 31, c = sl >> 2, s4 = sl & 3;
            lds_a[(c * PWS_PX + px) * 4 + (s4 ^ ((px >> 2) & 3))] = av[i];
        }
        uint4 wf[PWS_NS][2];
#pragma unroll
        for (int c = 0; c < PWS_NS; ++c)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf[c][ks] = wv[c][ks];
        __syncthreads();
        if (s + 1 < stages) fetch(s + 1);     // in flight while this stage's MFMAs run
        if (live) {
#pragma unroll
            for (int c = 0; c < PWS_NS; ++c)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) {
                        const int px = pt * 32 + r;
                        const uint4 xf = lds_a[(c * PWS_PX + px) * 4 + ((2 * ks + h) ^ ((px >> 2) & 3))];
                        acc[pt] = Tr<T>::mfma(wf[c][ks], xf, acc[pt]);
                    }
        }
    }
    if constexpr (MODE == PW_RESIDUAL) {
        // the turned epilogue of pointwise_mfma_kernel: a [32][36]-float slice per wave in the staging image (free once every wave is
        // past its last MFMA: one barrier, reached by the waves without a tile as well), 8 lanes own a pixel's 128 bytes
        __syncthreads();
        if (!live) return;
        float* slice = reinterpret_cast<float*>(lds_a) + wave * (32 * 36);
        const int piece = lane & 7, prow = lane >> 3;
        const int n = 32 * tile + 4 * piece;
        const f32x4 bs = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.chan_scale + n);
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const long mg = m0 + pt * 32;
            if (mg >= p.M) continue;
            f32x4 rs[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long m = mg + prow + 8 * i;
                rs[i] = m < p.M ? *reinterpret_cast<const f32x4*>(p.res_f32 + m * p.ldf + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[pt][4 * g], acc[pt][4 * g + 1], acc[pt][4 * g + 2], acc[pt][4 * g + 3]};
                *reinterpret_cast<f32x4*>(slice + r * 36 + 8 * g + 4 * h) = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long m = mg + prow + 8 * i;
                const f32x4 a = *reinterpret_cast<const f32x4*>(slice + (prow + 8 * i) * 36 + 4 * piece);
                f32x4 of;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float o = a[j] + bs[j];
                    of[j] = rs[i][j] + o * sc[j];
                }
                if (m < p.M) *reinterpret_cast<f32x4*>(p.out_f32 + m * p.ldf + n) = of;
            }
        }
    } else {
        if (!live) return;
        // typed output, 16 bytes per lane (v_permlane32_swap pairs the fragments of channel groups g, g + 1: see PW_STORE above)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const long m = m0 + pt * 32 + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint2 pk[2];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g = 2 * gp + gg;
                    const int n = 32 * tile + 8 * g + 4 * h;
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = acc[pt][4 * g + j] + (p.bias ? p.bias[n + j] : 0.f);
                    pk[gg] = pack4f<T>(o[0], o[1], o[2], o[3]);
                }
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                const u32x2_t sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                const u32x2_t sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out_typed) + m * p.ldo + 32 * tile + 16 * gp + 8 * h) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
        }
    }
}

// few pixels, a typed operand and nothing fused into the staging: the kernel above (FW_PW_SMALL_MAX_M=0 turns it off for A/B)
static bool pointwise_small_eligible(const PointwiseParams& p) {
    static const long max_m = [] {
        const char* e = getenv("FW_PW_SMALL_MAX_M");
        return e ? atol(e) : 16384L;
    }();
    if (p.M > max_m || p.a_f32 || p.gather2x2 || p.a_scale || p.ln_w || (p.lda & 7) || ((size_t)p.a & 15)) return false;
    // Many output channels: every 64-pixel workgroup column streams all of W again (N = 1152 / 2048 at 4096 pixels: 21.6 / 28.5 us here
    // against 23.4 / 25.5 in the 256-pixel kernel; N = 576 / 1024 at 16 k pixels: 23.6 / 37.2 against 17.7 / 22.1) - the kernel is for
    // N <= 384, which is every residual GEMM of the two levels (9.4 - 21.6 us against 21 - 55).  A 128-pixel form of it (half the weight
    // stream per pixel, 415 registers) for qkv / project_in was slower than the 256-pixel kernel as well: 10.53 against 10.33 ms per tile.
    if (p.N_tiles > 12) return false;
    if (p.mode == PW_RESIDUAL) return p.out_f32 && p.res_f32 && p.chan_scale && !(p.ldf & 3);
    return p.mode == PW_STORE && p.out_typed && !p.out_f32 && !(p.ldo & 7);
}

template <typename T>
static void launch_pw_small(const PointwiseParams& p, hipStream_t st) {
    const long mb = (p.M + PWS_PX - 1) / PWS_PX;
    const dim3 grid((unsigned)(mb * ((p.N_tiles + 3) / 4)));
    if (p.mode == PW_RESIDUAL)
        hipLaunchKernelGGL((pointwise_small_kernel<T, PW_RESIDUAL>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((pointwise_small_kernel<T, PW_STORE>), grid, dim3(256), 0, st, p);
    FW_HIP_CHECK(hipGetLastError());
}


template <typename T>
static void launch_pw_typed(const PointwiseParams& p, hipStream_t st) {
    const long mb = (p.M + PW_PX - 1) / PW_PX;
    dim3 block(256);
#define FW_PW(CT, MODE)                                                                                         \
    hipLaunchKernelGGL((pointwise_mfma_kernel<T, CT, MODE>), dim3((unsigned)(mb * ((p.N_tiles + CT - 1) / CT))), \
                       block, 0, st, p)
    switch (p.mode) {
        // (3 tiles = 96 output channels, Restormer's second level: one workgroup per pixel tile instead of two that both read it)
        case PW_STORE:
            if (p.N_tiles >= 4) FW_PW(4, PW_STORE); else if (p.N_tiles == 3) FW_PW(3, PW_STORE); else FW_PW(2, PW_STORE);
            break;
        case PW_RESIDUAL:
            if (p.N_tiles >= 4) FW_PW(4, PW_RESIDUAL); else if (p.N_tiles == 3) FW_PW(3, PW_RESIDUAL); else FW_PW(2, PW_RESIDUAL);
            break;
        case PW_SHUFFLE_UP:
            if (p.N_tiles >= 4) FW_PW(4, PW_SHUFFLE_UP); else FW_PW(2, PW_SHUFFLE_UP);
            break;
        case PW_GATE:
            // CT cout tiles per block = CT/2 gated tiles; N_tiles is the un-gated count (even)
            if (p.N_tiles >= 8) {
                hipLaunchKernelGGL((pointwise_mfma_kernel<T, 4, PW_GATE>), dim3((unsigned)(mb * (p.N_tiles / 4))), block, 0,
                                   st, p);
            } else {
                hipLaunchKernelGGL((pointwise_mfma_kernel<T, 2, PW_GATE>), dim3((unsigned)(mb * (p.N_tiles / 2))), block, 0,
                                   st, p);
            }
            break;
        default:
            throw Error(1, "pointwise: bad mode");
    }
#undef FW_PW
    FW_HIP_CHECK(hipGetLastError());
}

void launch_pointwise(DType dt, const PointwiseParams& p, hipStream_t st) {
    if (p.M <= 0 || p.K <= 0 || (p.K & 31) || p.N_tiles <= 0) throw Error(1, "pointwise: bad shape");
    if (pointwise_gemm_eligible(p)) {
        launch_pointwise_gemm(dt, p, st);
        return;
    }
    if (p.mode == PW_GATE && (p.N_tiles & 1)) throw Error(1, "pointwise: gate needs an even number of cout tiles");
    if (p.mode == PW_SHUFFLE_UP && (p.N_tiles & 3)) throw Error(1, "pointwise: PixelShuffle epilogue needs N % 128 == 0 (a 32-channel tile inside one sub-position)");
    if (p.gather2x2 && (p.Cin & 31)) throw Error(1, "pointwise: 2x2 gather needs Cin % 32 == 0");
    if (p.ln_w && (!p.ln_b || !p.a_f32 || p.gather2x2 || p.a_scale || p.K != 64 || p.lda != 64))
        throw Error(1, "pointwise: fused LayerNorm needs a plain fp32 [M][64] input");
    if (pointwise_small_eligible(p)) {
        if (dt == DT_BF16) launch_pw_small<__bf16>(p, st); else launch_pw_small<_Float16>(p, st);
        return;
    }
    if (dt == DT_BF16)
        launch_pw_typed<__bf16>(p, st);
    else
        launch_pw_typed<_Float16>(p, st);
}

// =====================================================================================================
// The second half of a NAFBlock at width 64 as ONE kernel (the full-resolution level, where every 1x1 conv is HBM-bound):
//   y   = inp + conv3(x * sca) * beta              x: the gated depthwise output (typed), inp: the fp32 stream
//   out = y + conv5(SimpleGate(conv4(norm2(y)))) * gamma
// Unfused, conv3 / conv4 (+LayerNorm, +gate) / conv5 move 640 + 384 + 640 B per pixel (the stream read three times and written
// twice, two typed intermediates); here the stream is read once and written once and x is read once: 640 B per pixel.  A
// pixel never leaves its lane between the three GEMMs: y stays in the accumulator registers (lane = pixel, 32 channels per
// 32x32 tile), LayerNorm reduces over the two lanes that share a pixel, and the typed operand of the next GEMM goes through a
// wave-private slice of the LDS image (a wave only ever reads the 64 pixel rows it wrote: no workgroup barrier inside the chain).
// Weights: the three packed matrices (32 KB) are copied to LDS once per workgroup; a workgroup walks a contiguous range of
// 256-pixel tiles.
// =====================================================================================================
struct NafTailParams {
    const void* x;          // typed [M][64]
    const float* a_scale;   // SCA factors [64]
    float* stream;          // fp32 [M][64], updated in place
    long M;
    const void *w3, *w4, *w5;                 // pack_pointwise_weights(64 x 64), (128 x 64), (64 x 64)
    const float *b3, *b4, *b5, *beta, *gamma, *ln_w, *ln_b;
    float ln_eps;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void naf_tail64_kernel(const NafTailParams p) {
    __shared__ __attribute__((aligned(16))) uint4 lds_a[2 * PW_PX * 4];        // two 32-channel chunk images of 256 pixels
    __shared__ __attribute__((aligned(16))) uint4 lds_w[(4 + 8 + 4) * 2 * 64]; // w3: [c][ks][2 tiles], w4: [c][ks][4], w5: [c][ks][2]
    // the per-channel vectors, read where they are used with one lane-constant LDS address + immediates (as global loads each
    // of the 64 (vector, group) pairs wanted its own 64-bit address register pair, hoisted out of the loops: 44 spills)
    __shared__ __attribute__((aligned(16))) float prm[9 * 64];   // b3, beta, ln_w, ln_b, b4 (128), b5, gamma, a_scale
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    {
        const float* src[8] = {p.b3, p.beta, p.ln_w, p.ln_b, p.b4, p.b5, p.gamma, p.a_scale};
        const int off[8] = {0, 64, 128, 192, 256, 384, 448, 512}, cnt[8] = {64, 64, 64, 64, 128, 64, 64, 64};
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (tid < cnt[k]) prm[off[k] + tid] = src[k][tid];
    }
    {   // weights -> LDS, fragment order as packed: [chunk][ks][tile][lane]
        const uint4* src[3] = {reinterpret_cast<const uint4*>(p.w3), reinterpret_cast<const uint4*>(p.w4), reinterpret_cast<const uint4*>(p.w5)};
        const int cnt[3] = {2 * 2 * 2 * 64, 2 * 2 * 4 * 64, 2 * 2 * 2 * 64}, base[3] = {0, 512, 1536};
#pragma unroll
        for (int k = 0; k < 3; ++k)
            for (int i = tid; i < cnt[k]; i += 256) lds_w[base[k] + i] = src[k][i];
    }
    __syncthreads();
    const uint4* W3 = lds_w;
    const uint4* W4 = lds_w + 512;
    const uint4* W5 = lds_w + 1536;
    const long tiles = (p.M + PW_PX - 1) / PW_PX;
    const long t_lo = blockIdx.x * tiles / gridDim.x, t_hi = (long)(blockIdx.x + 1) * tiles / gridDim.x;

    // the wave's B fragment of chunk c, k-step ks, pixel tile pt
    auto bfrag = [&](int c, int ks, int pt) {
        const int px = wave * 64 + pt * 32 + r;
        return lds_a[c * (PW_PX * 4) + px * 4 + ((2 * ks + h) ^ ((px >> 2) & 3))];
    };
    // lane's 4 channels 32 ct + 8 g + 4 h + {0..3} of pixel `px` as typed values into the image (wave-private rows)
    auto put4 = [&](int pt, int ct, int g, float a, float b, float c_, float d) {
        const int px = wave * 64 + pt * 32 + r;
        uint2* dst = reinterpret_cast<uint2*>(&lds_a[ct * (PW_PX * 4) + px * 4 + (g ^ ((px >> 2) & 3))]) + h;
        *dst = pack4f<T>(a, b, c_, d);
    };

    for (long t = t_lo; t < t_hi; ++t) {
        const long m0 = t * PW_PX;
        const float* a_scale = prm + 512;
        // ---- stage x (typed, scaled by the SCA factors) into the image: a wave stages its OWN 64 pixel rows, lane -> 16-byte slot
        //      lane & 3 of pixels (lane >> 2) + 16 i; the lane's 16 scale factors are the same for every piece -------------------
        float sc[2][8];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) sc[c][j] = a_scale[c * 32 + (lane & 3) * 8 + j];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int px = wave * 64 + (lane >> 2) + 16 * i, sl_ = lane & 3;
                const long m = m0 + px;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (m < p.M) {
                    v = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.x) + m * 64 + c * 32 + sl_ * 8);
                    float f[8];
                    unpack8f<T>(v, f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] *= sc[c][j];
                    v = pack8f<T>(f);
                }
                lds_a[c * (PW_PX * 4) + px * 4 + (sl_ ^ ((px >> 2) & 3))] = v;
            }
        // ---- the chain, one 32-pixel tile of the wave at a time (a rolled loop: both tiles at once need more than 256 registers) ---
#pragma unroll 1
        for (int pt = 0; pt < 2; ++pt) {
            const long m = m0 + wave * 64 + pt * 32 + r;
            // an offset the compiler cannot see through: the parameter reads stay where they are used (loop-invariant as they
            // are, it otherwise hoists all ~200 of them out of both loops and spills)
            int zz = 0;
            asm volatile("" : "+v"(zz));
            const float* pl = prm + zz;
            const float *b3 = pl, *beta = pl + 64, *ln_w = pl + 128, *ln_b = pl + 192, *b4 = pl + 256, *b5 = pl + 384, *gamma = pl + 448;
            // the stream row of the lane's pixel in accumulator layout: y[ct][4 g + j] = inp[px][32 ct + 8 g + 4 h + j]
            f32x16 y[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (m < p.M) v = *reinterpret_cast<const f32x4*>(p.stream + m * 64 + 32 * ct + 8 * g + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[ct][4 * g + j] = v[j];
                }
            // (the image rows a wave reads are rows it wrote itself: its own LDS writes are ordered before its reads)
            // conv3: y += (W3 . x + b3) * beta
            {
                f32x16 acc[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const uint4 xf = bfrag(c, ks, pt);
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) acc[ct] = Tr<T>::mfma(W3[((c * 2 + ks) * 2 + ct) * 64 + lane], xf, acc[ct]);
                    }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nn = 32 * ct + 8 * g + 4 * h;
                        const f32x4 bb = *reinterpret_cast<const f32x4*>(b3 + nn), be = *reinterpret_cast<const f32x4*>(beta + nn);
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[ct][4 * g + j] += (acc[ct][4 * g + j] + bb[j]) * be[j];
                    }
            }
            // norm2 over the pixel's 64 channels (32 in this lane, 32 in lane ^ 32), typed into the image
            {
                float sum = 0.f;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sum += y[ct][i];
                sum += __shfl_xor(sum, 32);
                const float mean = sum * (1.f / 64.f);
                float qq = 0.f;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float d = y[ct][i] - mean;
                        qq += d * d;
                    }
                qq += __shfl_xor(qq, 32);
                const float rstd = 1.0f / sqrtf(qq * (1.f / 64.f) + p.ln_eps);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nn = 32 * ct + 8 * g + 4 * h;
                        const f32x4 lw = *reinterpret_cast<const f32x4*>(ln_w + nn), lb = *reinterpret_cast<const f32x4*>(ln_b + nn);
                        float o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (y[ct][4 * g + j] - mean) * rstd * lw[j] + lb[j];
                        put4(pt, ct, g, o[0], o[1], o[2], o[3]);
                    }
            }
            // conv4 (128 outputs) + SimpleGate: g[n] = (a[n] + b4[n]) * (a[n + 64] + b4[n + 64]); the gated values overwrite the
            // normalised rows once the MFMAs have consumed them
            {
                f32x16 acc[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const uint4 xf = bfrag(c, ks, pt);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) acc[ct] = Tr<T>::mfma(W4[((c * 2 + ks) * 4 + ct) * 64 + lane], xf, acc[ct]);
                    }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nn = 32 * ct + 8 * g + 4 * h;
                        const f32x4 b1 = *reinterpret_cast<const f32x4*>(b4 + nn), b2 = *reinterpret_cast<const f32x4*>(b4 + nn + 64);
                        float o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (acc[ct][4 * g + j] + b1[j]) * (acc[ct + 2][4 * g + j] + b2[j]);
                        put4(pt, ct, g, o[0], o[1], o[2], o[3]);
                    }
            }
            // conv5 + gamma residual, stream out
            {
                f32x16 acc[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const uint4 xf = bfrag(c, ks, pt);
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) acc[ct] = Tr<T>::mfma(W5[((c * 2 + ks) * 2 + ct) * 64 + lane], xf, acc[ct]);
                    }
                if (m < p.M) {
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int nn = 32 * ct + 8 * g + 4 * h;
                            const f32x4 bb = *reinterpret_cast<const f32x4*>(b5 + nn), ga = *reinterpret_cast<const f32x4*>(gamma + nn);
                            f32x4 o;
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = y[ct][4 * g + j] + (acc[ct][4 * g + j] + bb[j]) * ga[j];
                            *reinterpret_cast<f32x4*>(p.stream + m * 64 + nn) = o;
                        }
                }
            }
        }
    }
}

void launch_naf_tail64(DType dt, const void* x, const float* a_scale, float* stream, long M, const void* w3, const float* b3,
                       const float* beta, const float* ln_w, const float* ln_b, float ln_eps, const void* w4, const float* b4,
                       const void* w5, const float* b5, const float* gamma, hipStream_t st) {
    NafTailParams p{x, a_scale, stream, M, w3, w4, w5, b3, b4, b5, beta, gamma, ln_w, ln_b, ln_eps};
    const long tiles = (M + PW_PX - 1) / PW_PX;
    // two workgroups per CU are resident (64 KB of LDS each): a grid of that many, each walking a contiguous range of tiles
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long grid = tiles < 2L * cus ? tiles : 2L * cus;
    if (dt == DT_BF16)
        hipLaunchKernelGGL((naf_tail64_kernel<__bf16>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((naf_tail64_kernel<_Float16>), dim3((unsigned)grid), dim3(256), 0, st, p);
    FW_HIP_CHECK(hipGetLastError());
}

// Host-side packer for pointwise weights: w[cout][K] fp32 (already in the kernel's k order) ->
// fragments [chunk][ks][cout tile][lane][8]; returns uint16 count.
size_t pack_pointwise_weights(DType dt, const float* w, int cout, int K, uint16_t* dst) {
    const int nt = (cout + 31) / 32, chunks = (K + 31) / 32;
    const size_t n = (size_t)chunks * 2 * nt * 64 * 8;
    if (!dst) return n;
    size_t o = 0;
    for (int c = 0; c < chunks; ++c)
        for (int ks = 0; ks < 2; ++ks)
            for (int t = 0; t < nt; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = 32 * t + (lane & 31);
                        const int k = 32 * c + 16 * ks + 8 * (lane >> 5) + j;
                        dst[o++] = f32_to_operand(dt, (co < cout && k < K) ? w[(size_t)co * K + k] : 0.f);
                    }
    return n;
}

// =====================================================================================================
// LayerNorm2d over channels, fp32 in -> operand-typed out.  LPP lanes per pixel (16 / 32 / 64 for C <= 64 / 128 / more), so
// a wave normalises 64 / LPP pixels at once and every lane is busy at the narrow full-resolution levels (the first version
// spent a whole wave on a pixel: at width 64, where four fifths of the LayerNorm bytes are, 48 of its 64 lanes idled).
// Lane l of a pixel's group holds channels 4l..4l+3 (+256 per extra pass); the xor-shuffle tree over the group adds the
// same terms in the same order as the 64-lane tree did with its idle lanes at zero.  eps = 1e-6.
// =====================================================================================================
template <typename T, int LPP>
__global__ __launch_bounds__(256) void layernorm2d_kernel(const float* __restrict__ x, long M, int C, const float* w,
                                                          const float* b, T* out, float eps) {
    constexpr int PPW = 64 / LPP;  // pixels per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPP, cl = lane % LPP;
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const long groups = (M + PPW - 1) / PPW;
    for (long gi = wave0; gi < groups; gi += nwaves) {
        const long m = gi * PPW + sub;
        const bool live = m < M;
        const float* row = x + (live ? m : 0) * C;
        float v[16];  // up to C = 1024
        float s = 0.f;
        const int passes = (C + 255) / 256;
#pragma unroll
        for (int pss = 0; pss < (LPP == 64 ? 4 : 1); ++pss) {
            if (pss < passes) {
                const int c0 = pss * 256 + cl * 4;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (c0 < C && live) t = *reinterpret_cast<const f32x4*>(row + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[pss * 4 + j] = t[j];
                    s += t[j];
                }
            }
        }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / C;
        float q = 0.f;
#pragma unroll
        for (int pss = 0; pss < (LPP == 64 ? 4 : 1); ++pss) {
            if (pss < passes) {
                const int c0 = pss * 256 + cl * 4;
                if (c0 < C) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = v[pss * 4 + j] - mean;
                        q += d * d;
                    }
                }
            }
        }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / C + eps);
#pragma unroll
        for (int pss = 0; pss < (LPP == 64 ? 4 : 1); ++pss) {
            if (pss < passes) {
                const int c0 = pss * 256 + cl * 4;
                if (c0 < C && live) {
                    float o4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o4[j] = (v[pss * 4 + j] - mean) * rstd * w[c0 + j] + b[c0 + j];
                    *reinterpret_cast<uint2*>(out + m * C + c0) = pack4f<T>(o4[0], o4[1], o4[2], o4[3]);
                }
            }
        }
    }
}

void launch_layernorm2d(DType dt, const float* x, long M, int C, const float* w, const float* b, void* out,
                        hipStream_t st) {
    if (C < 4 || (C & 3) || C > 1024) throw Error(1, "layernorm2d: C must be a multiple of 4, <= 1024");
    const int lpp = C <= 64 ? 16 : (C <= 128 ? 32 : 64);
    const long groups = (M + 64 / lpp - 1) / (64 / lpp);
    const long blocks = (groups + 3) / 4 < 4096 ? (groups + 3) / 4 : 4096;
#define FW_LN(T, L) hipLaunchKernelGGL((layernorm2d_kernel<T, L>), dim3((unsigned)blocks), dim3(256), 0, st, x, M, C, w, b, (T*)out, 1e-6f)
    if (dt == DT_BF16) {
        if (lpp == 16) FW_LN(__bf16, 16); else if (lpp == 32) FW_LN(__bf16, 32); else FW_LN(__bf16, 64);
    } else {
        if (lpp == 16) FW_LN(_Float16, 16); else if (lpp == 32) FW_LN(_Float16, 32); else FW_LN(_Float16, 64);
    }
#undef FW_LN
    FW_HIP_CHECK(hipGetLastError());
}

// =====================================================================================================
// Depthwise 3x3 (zero pad 1) on a 2C-channel operand-typed NHWC tensor, fused SimpleGate (out[c] = dw[c] * dw[c+C]) and
// per-channel partial sums of the gated output (for the SCA global average pool).  One thread = 8 gated channels of a column
// of GATE_ROWS output pixels.
// =====================================================================================================
constexpr int DW_MAX_BLOCKS = 768;   // 256 CUs x the 3 blocks that fit on one (168 VGPRs): a 4th round at a third of the occupancy was half the kernel
constexpr int GATE_ROWS = 3;

template <typename T>
__global__ __launch_bounds__(256, 3) void dwconv3x3_gate_kernel(const T* __restrict__ x, int H, int W, int C, const float* wdw,
                                                                const float* bdw, T* out, float* partial) {
    // wdw: [2C][9] fp32, bdw: [2C].  groups = C/8 divides 256 (C is a power-of-two multiple of 32 here), so a thread keeps
    // the same channel group for its whole grid-stride loop and the SCA pooling is a fixed-order (deterministic) reduction:
    // registers -> LDS -> partial[block][C] -> sca_kernel.
    // The filters sit in LDS transposed to [tap][2C] (dynamic shared memory, 72*C bytes).  Per item a thread loads the
    // GATE_ROWS + 2 input rows x 3 pixels of each gate half with 16-byte loads that have no branch between them (addresses
    // clamped into the image, values masked to zero outside), every load feeds up to three output rows, and the half's 72
    // taps sit in registers for the whole column.  (Before: one pixel per thread, 18 loads each behind its own border test
    // and its own wait, two 64-bit divisions per pixel, the filters re-read from LDS for every pixel.)
    extern __shared__ __attribute__((aligned(16))) float dw_smem[];
    float* wl = dw_smem;                                     // [9][2C]
    float (*red)[8] = reinterpret_cast<float (*)[8]>(dw_smem + 18 * C);  // [256][8]
    for (int i = threadIdx.x; i < 18 * C; i += 256) {
        const int ch = i / 9, tap = i - ch * 9;
        wl[tap * 2 * C + ch] = wdw[i];
    }
    __syncthreads();
    const unsigned groups = C / 8;
    const unsigned strips = (H + GATE_ROWS - 1) / GATE_ROWS;
    const unsigned total = strips * (unsigned)W * groups;    // (strip, x, group), group fastest
    const int g = threadIdx.x % groups;
    float cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    using V8 = typename Tr<T>::v8;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned col = idx / groups;                   // strip * W + x
        const int strip = (int)(col / (unsigned)W), xx = (int)(col - (unsigned)strip * W);
        const int y0 = strip * GATE_ROWS;
        float res[GATE_ROWS][8];
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int c0 = half * C + g * 8;
            float acc[GATE_ROWS][8];
#pragma unroll
            for (int o = 0; o < GATE_ROWS; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[o][j] = bdw[c0 + j];
            // one tap column at a time: its GATE_ROWS + 2 input pixels and its three filter rows are all that is live
            // (24 weights instead of 72: twice the waves per SIMD)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int sx = xx + dx - 1;
                const int cx = sx < 0 ? 0 : (sx >= W ? W - 1 : sx);
                const bool cok = sx >= 0 && sx < W;
                V8 f[GATE_ROWS + 2];
#pragma unroll
                for (int r = 0; r < GATE_ROWS + 2; ++r) {
                    const int sy = y0 + r - 1;
                    const int cy = sy < 0 ? 0 : (sy >= H ? H - 1 : sy);
                    f[r] = __builtin_bit_cast(V8, *reinterpret_cast<const uint4*>(x + ((long)cy * W + cx) * (2 * C) + c0));
                }
                float wr[3][8];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + (dy * 3 + dx) * 2 * C + c0);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(wl + (dy * 3 + dx) * 2 * C + c0 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        wr[dy][j] = w0[j];
                        wr[dy][4 + j] = w1[j];
                    }
                }
#pragma unroll
                for (int r = 0; r < GATE_ROWS + 2; ++r) {
                    const int sy = y0 + r - 1;
                    const float m = (cok && sy >= 0 && sy < H) ? 1.f : 0.f;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (float)f[r][j] * m;
#pragma unroll
                    for (int o = 0; o < GATE_ROWS; ++o) {
                        const int dy = r - o;
                        if (dy >= 0 && dy < 3) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[o][j] += v[j] * wr[dy][j];
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < GATE_ROWS; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) res[o][j] = half ? res[o][j] * acc[o][j] : acc[o][j];
        }
#pragma unroll
        for (int o = 0; o < GATE_ROWS; ++o) {
            if (y0 + o >= H) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) cs[j] += res[o][j];
            *reinterpret_cast<uint4*>(out + ((long)(y0 + o) * W + xx) * C + g * 8) = pack8f<T>(res[o]);
        }
    }
    if (partial) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = cs[j];
        __syncthreads();
        if (threadIdx.x < C) {
            const int gg = threadIdx.x / 8, j = threadIdx.x % 8;
            float s = 0.f;
            for (int t = gg; t < 256; t += groups) s += red[t][j];
            partial[(long)blockIdx.x * C + threadIdx.x] = s;
        }
        if (C > 256) {
            for (int c = 256 + threadIdx.x; c < C; c += 256) {
                const int gg = c / 8, j = c % 8;
                float s = 0.f;
                for (int t = gg; t < 256; t += groups) s += red[t][j];
                partial[(long)blockIdx.x * C + c] = s;
            }
        }
    }
}

static bool dwconv_wide() {   // FW_NAF_DW_WIDE=0: the one-kernel-for-all-widths path (A/B)
    static const bool on = [] {
        const char* e = getenv("FW_NAF_DW_WIDE");
        return !e || atoi(e) != 0;
    }();
    return on;
}

// The same kernel for C >= 256 (the low-resolution levels: 8 k ... 130 k pixels, 512 ... 2048 channels).  There every block of the
// kernel above staged ALL 2C filters (72 KB at C = 1024, 768 blocks: 1.0 GB of the 5.6 GB the 28 launches of a forward moved, PMC)
// because its 256 threads spanned every channel group.  Here a block owns 32 channel groups (256 gated channels: 18 KB of filters)
// and walks columns; blocks b, b + 1, .. b + C/256 - 1 cover the channel ranges of one column set and write disjoint slices of one
// `partial` row, so the SCA pooling stays a fixed-order sum over rows.
template <typename T>
__global__ __launch_bounds__(256, 3) void dwconv3x3_gate_wide_kernel(const T* __restrict__ x, int H, int W, int C, const float* wdw,
                                                                     const float* bdw, T* out, float* partial) {
    __shared__ __attribute__((aligned(16))) float wl[9 * 512];      // [tap][half][256 channels of this block's range]
    __shared__ float red[256][8];
    const int ngr = C / 256;                                        // channel ranges
    const int gr = blockIdx.x % ngr, cb = blockIdx.x / ngr, ncb = gridDim.x / ngr;
    for (int i = threadIdx.x; i < 9 * 512; i += 256) {
        const int tap = i / 512, r = i - tap * 512, half = r >> 8;
        wl[i] = wdw[(size_t)(half * C + gr * 256 + (r & 255)) * 9 + tap];
    }
    __syncthreads();
    const int gl = threadIdx.x & 31, cl = threadIdx.x >> 5;         // channel group of the range, column of the block's eight
    const int g = gr * 32 + gl;
    const unsigned strips = (H + GATE_ROWS - 1) / GATE_ROWS;
    const unsigned cols = strips * (unsigned)W;                     // (strip, x)
    float cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    using V8 = typename Tr<T>::v8;
    for (unsigned col = cb * 8 + cl; col < cols; col += ncb * 8) {
        const int strip = (int)(col / (unsigned)W), xx = (int)(col - (unsigned)strip * W);
        const int y0 = strip * GATE_ROWS;
        float res[GATE_ROWS][8];
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int c0 = half * C + g * 8;
            const float* wh = wl + half * 256 + gl * 8;
            float acc[GATE_ROWS][8];
#pragma unroll
            for (int o = 0; o < GATE_ROWS; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[o][j] = bdw[c0 + j];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int sx = xx + dx - 1;
                const int cx = sx < 0 ? 0 : (sx >= W ? W - 1 : sx);
                const bool cok = sx >= 0 && sx < W;
                V8 f[GATE_ROWS + 2];
#pragma unroll
                for (int r = 0; r < GATE_ROWS + 2; ++r) {
                    const int sy = y0 + r - 1;
                    const int cy = sy < 0 ? 0 : (sy >= H ? H - 1 : sy);
                    f[r] = __builtin_bit_cast(V8, *reinterpret_cast<const uint4*>(x + ((long)cy * W + cx) * (2 * C) + c0));
                }
                float wr[3][8];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wh + (dy * 3 + dx) * 512);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(wh + (dy * 3 + dx) * 512 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        wr[dy][j] = w0[j];
                        wr[dy][4 + j] = w1[j];
                    }
                }
#pragma unroll
                for (int r = 0; r < GATE_ROWS + 2; ++r) {
                    const int sy = y0 + r - 1;
                    const float m = (cok && sy >= 0 && sy < H) ? 1.f : 0.f;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (float)f[r][j] * m;
#pragma unroll
                    for (int o = 0; o < GATE_ROWS; ++o) {
                        const int dy = r - o;
                        if (dy >= 0 && dy < 3) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[o][j] += v[j] * wr[dy][j];
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < GATE_ROWS; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) res[o][j] = half ? res[o][j] * acc[o][j] : acc[o][j];
        }
#pragma unroll
        for (int o = 0; o < GATE_ROWS; ++o) {
            if (y0 + o >= H) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) cs[j] += res[o][j];
            *reinterpret_cast<uint4*>(out + ((long)(y0 + o) * W + xx) * C + g * 8) = pack8f<T>(res[o]);
        }
    }
    if (partial) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = cs[j];
        __syncthreads();
        // channel c of the range = group c / 8, element c % 8: the eight threads gl, gl + 32, ... hold its column sums
        const int gg = threadIdx.x >> 3, j = threadIdx.x & 7;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) sum += red[gg + 32 * t][j];
        partial[(long)cb * C + gr * 256 + threadIdx.x] = sum;
    }
}

// rows of `partial` the depthwise kernel writes (= the workgroups that cover one channel range)
static int dwconv_wide_rows(int H, int W, int C) {
    const long cols = (long)((H + GATE_ROWS - 1) / GATE_ROWS) * W;
    const long want = (cols + 7) / 8;
    const long cap = DW_MAX_BLOCKS / (C / 256);
    return (int)(want < cap ? want : cap);
}

int dwconv_blocks(int H, int W, int C) {
    if (C >= 256 && dwconv_wide()) return dwconv_wide_rows(H, W, C);
    const long total = (long)((H + GATE_ROWS - 1) / GATE_ROWS) * W * (C / 8);
    const long b = (total + 255) / 256;
    return (int)(b < DW_MAX_BLOCKS ? b : DW_MAX_BLOCKS);
}

void launch_dwconv3x3_gate(DType dt, const void* x, int H, int W, int C, const float* wdw, const float* bdw, void* out,
                           float* partial, hipStream_t st) {
    if (C < 32 || (C & (C - 1)) || C > 1024) throw Error(1, "dwconv3x3_gate: C must be a power of two in [32, 1024]");
    if (C >= 256 && dwconv_wide()) {
        const int grid = dwconv_wide_rows(H, W, C) * (C / 256);
        if (dt == DT_BF16)
            hipLaunchKernelGGL((dwconv3x3_gate_wide_kernel<__bf16>), dim3((unsigned)grid), dim3(256), 0, st, (const __bf16*)x, H, W, C, wdw, bdw,
                               (__bf16*)out, partial);
        else
            hipLaunchKernelGGL((dwconv3x3_gate_wide_kernel<_Float16>), dim3((unsigned)grid), dim3(256), 0, st, (const _Float16*)x, H, W, C, wdw, bdw,
                               (_Float16*)out, partial);
        FW_HIP_CHECK(hipGetLastError());
        return;
    }
    const int blocks = dwconv_blocks(H, W, C);
    const size_t smem = ((size_t)18 * C + 256 * 8) * sizeof(float);  // filters [9][2C] + pooling scratch (<= 80 KiB)
    static const bool attr_set = [] {
        const int cap = (18 * 1024 + 256 * 8) * (int)sizeof(float);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_gate_kernel<__bf16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_gate_kernel<_Float16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        return true;
    }();
    (void)attr_set;
    if (dt == DT_BF16)
        hipLaunchKernelGGL((dwconv3x3_gate_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), smem, st, (const __bf16*)x, H,
                           W, C, wdw, bdw, (__bf16*)out, partial);
    else
        hipLaunchKernelGGL((dwconv3x3_gate_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), smem, st,
                           (const _Float16*)x, H, W, C, wdw, bdw, (_Float16*)out, partial);
    FW_HIP_CHECK(hipGetLastError());
}

// SCA: mean[k] = (sum over blocks of partial[b][k]) / HW, summed in block order (deterministic); then
// s[n] = b[n] + sum_k W[n][k] * mean[k], one wave per output channel.
__global__ __launch_bounds__(1024) void sca_mean_kernel(const float* __restrict__ partial, int nblocks, float inv_hw, int C,
                                                        float* mean) {
    // workgroup = 64 channels x 16 slices of the block list; slice s sums blocks s, s+16, ... in order, then one thread per
    // channel adds the 16 slice sums in order: a fixed summation tree, deterministic
    __shared__ float part[16][64];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + c;
    float m = 0.f;
    if (k < C)
        for (int q = sl; q < nblocks; q += 16) m += partial[(long)q * C + k];
    part[sl][c] = m;
    __syncthreads();
    if (sl == 0 && k < C) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][c];
        mean[k] = t * inv_hw;
    }
}

__global__ __launch_bounds__(256) void sca_kernel(const float* __restrict__ mean, int C, const float* w, const float* b,
                                                  float* s) {
    const int lane = threadIdx.x & 63;
    const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (n >= C) return;
    float a = 0.f;
    for (int k = lane; k < C; k += 64) a += w[(long)n * C + k] * mean[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) s[n] = a + b[n];
}

// `s` must have room for 2*C floats: [0, C) receives the scale, [C, 2C) is scratch for the pooled mean.
void launch_sca(const float* partial, int nblocks, long HW, int C, const float* w, const float* b, float* s,
                hipStream_t st) {
    float* mean = s + C;
    hipLaunchKernelGGL(sca_mean_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, partial, nblocks, 1.0f / (float)HW, C, mean);
    hipLaunchKernelGGL(sca_kernel, dim3((C * 64 + 255) / 256), dim3(256), 0, st, (const float*)mean, C, w, b, s);
    FW_HIP_CHECK(hipGetLastError());
}

// fp32 NHWC [M][C] -> operand-typed chunk-planar [C/32][M][32] (input layout of conv3x3_mfma's trunk form)
template <typename T>
__global__ __launch_bounds__(256) void f32_to_planar_kernel(const float* __restrict__ x, long M, int C, T* out) {
    const int groups = C / 8;
    const long total = M * groups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / groups;
        const int g = (int)(i - m * groups);
        const f32x4* src = reinterpret_cast<const f32x4*>(x + m * C + g * 8);
        const f32x4 lo = src[0], hi = src[1];
        float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const int plane = g >> 2, s = g & 3;
        *reinterpret_cast<uint4*>(out + ((size_t)plane * M + m) * 32 + s * 8) = pack8f<T>(f);
    }
}

void launch_f32_to_planar(DType dt, const float* x, long M, int C, void* out, hipStream_t st) {
    if (C & 31) throw Error(1, "f32_to_planar: C must be a multiple of 32");
    const long total = M * (C / 8);
    const long blocks = (total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096;
    if (dt == DT_BF16)
        hipLaunchKernelGGL((f32_to_planar_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, st, x, M, C, (__bf16*)out);
    else
        hipLaunchKernelGGL((f32_to_planar_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, st, x, M, C,
                           (_Float16*)out);
    FW_HIP_CHECK(hipGetLastError());
}

}  // namespace fw
