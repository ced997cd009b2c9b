// LayerNorm -> 1x1 convolution -> depthwise 3x3 -> gate in one kernel: the first half of a NAFBlock at the wide levels
// (reference: the NAFNet the TAP driver loads, tap_denoise.py:299-333; block layout as in oracle/nafnet_ref.py).
//
// Unfused, the 2c-channel tensor between conv1 and the depthwise conv is written and read once each (512 of the 1536 bytes a
// width-64 block moved per pixel) and the depthwise kernel itself ran at 2.1 TB/s (three overlapping 16-byte gathers per
// tap column through the texture path).  Here that tensor only exists in LDS:
//
//   * a persistent 512-thread workgroup walks 14 x 30-pixel output tiles; conv1 is evaluated on the 16 x 32 halo tile
//     (x1.22 MACs - they are cheap: K = c);
//   * phase A: wave w loads the fp32 stream of halo rows 2w, 2w+1 straight into registers (lane = pixel l & 15, quarter
//     l >> 4 of every 16-channel group: each load instruction covers 64 contiguous bytes per pixel), LayerNorm2d statistics
//     with two xor-shuffles, (x - mean) * rstd becomes the B fragments of v_mfma_f32_16x16x32 (c / 2 registers); the affine
//     part of the LayerNorm is folded into conv1 by the host (W' = W diag(gamma), b' = b + W beta: pack_pw_dw_gate_weights);
//   * per chunk of 64 conv1 channels (32 x1 channels and the 32 x2 channels they are gated with): the GEMM from LDS-resident
//     weight fragments into 16 accumulator tiles, + bias, zero outside the image (the depthwise conv pads conv1's OUTPUT),
//     typed, into a [512 px][136 B] LDS image (stride 34 dwords: 16 lanes of consecutive pixels cover all 32 banks);
//   * depthwise 3x3 + SimpleGate from LDS: wave w owns channels 4w..4w+3 of the chunk's 32 pairs, so its 72 + 8 filter taps
//     and biases are wave-uniform (SGPRs); lane = (7-row strip, column); 27 ds_read_b64 per half feed 63 x 2 v_pk_fma_f32;
//     the gated pixels go to HBM as 8-byte stores, their sums stay in registers for the SCA pooling
//     (wave reduction at the end of the kernel -> partial[workgroup][c], the fixed-order scheme of dwconv3x3_gate_kernel).
#include "fw_internal.h"
#include "conv_common.h"

namespace fw {

constexpr int FR_HR = 16, FR_HC = 32;                 // halo tile
constexpr int FR_OR = 14, FR_OC = 30;                 // output tile
constexpr int FR_PXB = 136;                           // LDS bytes per pixel record: 64 channels + 8 pad
constexpr int FR_Y_BYTES = FR_HR * FR_HC * FR_PXB;    // 69632
constexpr int FR_STRIP = 7;                           // output rows per depthwise item

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T, int CIN>
__global__ __launch_bounds__(512, 2) void pw_dw_gate_kernel(const PwDwParams p) {
    constexpr int KC = CIN / 32;                      // 32-channel chunks of the contraction
    constexpr int NCH = 2 * CIN / 64;                 // 64-channel chunks of conv1's output
    constexpr int W_BYTES = NCH * KC * 4 * 1024;
    __shared__ __attribute__((aligned(16))) char ybuf[FR_Y_BYTES];
    __shared__ __attribute__((aligned(16))) uint4 wl[W_BYTES / 16];
    __shared__ __attribute__((aligned(16))) float dwl[11 * 2 * CIN];   // depthwise filters [9][2c] + bias [2c], conv1 bias [2c]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, q = lane & 15, sl = lane >> 4;

    for (int i = tid; i < W_BYTES / 16; i += 512) wl[i] = reinterpret_cast<const uint4*>(p.wpk)[i];
    for (int i = tid; i < 11 * 2 * CIN; i += 512)
        dwl[i] = i < 9 * 2 * CIN ? p.wdw_t[i] : (i < 10 * 2 * CIN ? p.bdw[i - 9 * 2 * CIN] : p.bias[i - 10 * 2 * CIN]);
    __syncthreads();

    const int tiles_x = (p.W + FR_OC - 1) / FR_OC, tiles_y = (p.H + FR_OR - 1) / FR_OR;
    const long ntiles = (long)tiles_x * tiles_y;
    const long t_lo = blockIdx.x * ntiles / gridDim.x, t_hi = (long)(blockIdx.x + 1) * ntiles / gridDim.x;

    // depthwise item of this lane: strip (0/1) and column; lanes 60-63 idle
    const bool dw_on = lane < 2 * FR_OC;
    const int strip = lane >= FR_OC ? 1 : 0;
    const int col = dw_on ? lane - strip * FR_OC : 0;
    const char* yrd = ybuf + ((FR_STRIP * strip) * FR_HC + col) * FR_PXB + 8 * wave;

    f32x4 cs[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) cs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef FW_FRONT_STAMP   // diagnostic build: cycles per phase of one wave, printed at the end
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = __builtin_amdgcn_s_memtime(), t_begin = tq;
#define FW_PH(i)                                                  \
    do {                                                          \
        const unsigned long long now = __builtin_amdgcn_s_memtime(); \
        ph[i] += now - tq;                                        \
        tq = now;                                                 \
    } while (0)
#else
#define FW_PH(i)
#endif
    const float inv_c = 1.0f / (float)CIN;
    const int C = CIN;

    for (long tile = t_lo; tile < t_hi; ++tile) {
        const int tyi = (int)(tile / tiles_x);
        const int ty0 = tyi * FR_OR, tx0 = (int)(tile - (long)tyi * tiles_x) * FR_OC;

        // ---- phase A: this wave's 64 halo pixels -> normalised B fragments ------------------------------------------------
        uint4 xb[4][KC];
        unsigned inside = 0;
        constexpr int TB = CIN > 64 ? 2 : 4;          // pixel tiles whose raw fp32 is in flight together (64 registers)
#pragma unroll
        for (int t0 = 0; t0 < 4; t0 += TB) {
            f32x4 v[TB][2 * KC];
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) {
                const int t = t0 + tt;
                const int gy = ty0 - 1 + 2 * wave + (t >> 1), gx = tx0 - 1 + 16 * (t & 1) + q;
                inside |= (unsigned)(gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) << t;
                const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
#ifdef FW_FRONT_ABL_LOAD   // timing only: every tile reads the first tile's pixels (L2 hits)
                const float* src = p.x + ((long)(2 * wave + (t >> 1)) * p.W + 16 * (t & 1) + q) * p.ldx + 4 * sl;
#else
                const float* src = p.x + ((long)cy * p.W + cx) * p.ldx + 4 * sl;
#endif
#pragma unroll
                for (int m = 0; m < 2 * KC; ++m) v[tt][m] = *reinterpret_cast<const f32x4*>(src + 16 * m);
            }
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) {
                const int t = t0 + tt;
                float s = 0.f;
#pragma unroll
                for (int m = 0; m < 2 * KC; ++m) s += (v[tt][m][0] + v[tt][m][1]) + (v[tt][m][2] + v[tt][m][3]);
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                const float mean = s * inv_c;
                float ss = 0.f;
#pragma unroll
                for (int m = 0; m < 2 * KC; ++m) {
                    v[tt][m] = v[tt][m] - mean;
                    ss += (v[tt][m][0] * v[tt][m][0] + v[tt][m][1] * v[tt][m][1]) + (v[tt][m][2] * v[tt][m][2] + v[tt][m][3] * v[tt][m][3]);
                }
                ss += __shfl_xor(ss, 16);
                ss += __shfl_xor(ss, 32);
                const float rstd = 1.0f / __builtin_sqrtf(ss * inv_c + p.ln_eps);
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    uint2 h[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const f32x4 n = v[tt][2 * kc + u] * rstd;
                        h[u] = Op<T>::pack4(n[0], n[1], n[2], n[3]);
                    }
                    xb[t][kc] = make_uint4(h[0].x, h[0].y, h[1].x, h[1].y);
                }
            }
        }

        FW_PH(0);
#pragma unroll 1
        for (int j = 0; j < NCH; ++j) {
            // ---- conv1, chunk j: 64 output channels x this wave's 64 pixels ------------------------------------------------
            f32x4 acc[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const uint4 wf = wl[((j * KC + kc) * 4 + ct) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t][ct] = Op<T>::mfma16(wf, xb[t][kc], acc[t][ct]);
                }
            FW_PH(1);
            __syncthreads();   // the depthwise pass over the previous chunk is done with ybuf
            FW_PH(2);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int cc = 16 * ct + 4 * sl;                                   // channel of the chunk
                const int co = cc < 32 ? 32 * j + cc : C + 32 * j + (cc - 32);     // conv1 output channel
                const f32x4 bs = *reinterpret_cast<const f32x4*>(dwl + 10 * 2 * C + co);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x4 y = acc[t][ct] + bs;
                    if (!((inside >> t) & 1u)) y = f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<uint2*>(ybuf + (64 * wave + 16 * t + q) * FR_PXB + 2 * cc) = Op<T>::pack4(y[0], y[1], y[2], y[3]);
                }
            }
            FW_PH(3);
            __syncthreads();
            FW_PH(4);

            // ---- depthwise 3x3 + SimpleGate: channels 32 j + 4 wave .. + 3 (x1) and C + the same (x2) -----------------------
            f32x2 x1a[FR_STRIP], x1b[FR_STRIP];
            f32x4 cj = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
#pragma clang fp contract(fast)
                const int ch0 = half * C + 32 * j + 4 * wave;
                f32x2 wa[9], wb[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(dwl + k * 2 * C + ch0);
                    wa[k] = f32x2{wv[0], wv[1]};
                    wb[k] = f32x2{wv[2], wv[3]};
                }
                const f32x4 bv = *reinterpret_cast<const f32x4*>(dwl + 9 * 2 * C + ch0);
                const f32x2 ba = {bv[0], bv[1]}, bb = {bv[2], bv[3]};
                f32x2 aa[FR_STRIP], ab[FR_STRIP];
#pragma unroll
                for (int o = 0; o < FR_STRIP; ++o) {
                    aa[o] = ba;
                    ab[o] = bb;
                }
                // one input row ahead of the FMAs; the scheduling barriers keep hipcc from hoisting all 27 reads (54 registers)
                const char* yh = yrd + 64 * half;
                uint2 cur[3], nxt[3];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) cur[dx] = *reinterpret_cast<const uint2*>(yh + dx * FR_PXB);
#ifdef FW_FRONT_ABL_DW     // timing only: one input row instead of nine
                constexpr int NR = 1;
#else
                constexpr int NR = FR_STRIP + 2;
#endif
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if (r + 1 < FR_STRIP + 2) {
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) nxt[dx] = *reinterpret_cast<const uint2*>(yh + ((r + 1) * FR_HC + dx) * FR_PXB);
                    }
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const f32x4 f = Op<T>::unpack4(cur[dx]);
                        const f32x2 fa = {f[0], f[1]}, fb = {f[2], f[3]};
#pragma unroll
                        for (int o = 0; o < FR_STRIP; ++o) {
                            const int dy = r - o;
                            if (dy >= 0 && dy < 3) {
                                aa[o] = fa * wa[dy * 3 + dx] + aa[o];
                                ab[o] = fb * wb[dy * 3 + dx] + ab[o];
                            }
                        }
                    }
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) cur[dx] = nxt[dx];
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (half == 0) {
#pragma unroll
                    for (int o = 0; o < FR_STRIP; ++o) {
                        x1a[o] = aa[o];
                        x1b[o] = ab[o];
                    }
                } else {
                    // The gated pixel overwrites its own x1 slot in the LDS image (this wave's channels: nobody else reads them, and
                    // the wave's own reads are all issued) and leaves in 64-byte pieces below: eight waves x 8 bytes per pixel straight
                    // from registers were 16 partial writes per 128-byte line (21 % of the kernel, FW_FRONT_DIRECT_STORE to compare).
                    const int gx = tx0 + col;
#ifdef FW_FRONT_DIRECT_STORE
                    T* orow = reinterpret_cast<T*>(p.out) + 32 * j + 4 * wave;
#endif
#pragma unroll
                    for (int o = 0; o < FR_STRIP; ++o) {
                        const int gy = ty0 + FR_STRIP * strip + o;
                        const f32x2 ga = x1a[o] * aa[o], gb = x1b[o] * ab[o];
                        const uint2 gp = Op<T>::pack4(ga[0], ga[1], gb[0], gb[1]);
                        if (dw_on) {
#ifndef FW_FRONT_DIRECT_STORE
                            *reinterpret_cast<uint2*>(const_cast<char*>(yrd) + ((o + 1) * FR_HC + 1) * FR_PXB) = gp;
#endif
                            if (gy < p.H && gx < p.W) {
#if defined(FW_FRONT_DIRECT_STORE) && !defined(FW_FRONT_ABL_STORE)
                                *reinterpret_cast<uint2*>(orow + ((long)gy * p.W + gx) * p.ldo) = gp;
#endif
                                cj += f32x4{ga[0], ga[1], gb[0], gb[1]};
                            }
                        }
                    }
                }
            }
            FW_PH(5);
#ifndef FW_FRONT_DIRECT_STORE
            __syncthreads();
            FW_PH(6);
#ifndef FW_FRONT_ABL_STORE   // timing only: no stores (the sums keep the arithmetic alive)
#pragma unroll
            for (int k = 0; k < (FR_OR * FR_OC * 4 + 511) / 512; ++k) {
                const int i = tid + 512 * k;                  // (output pixel, 16-byte quarter of its 32 gated channels)
                const int px = i >> 2, part = i & 3;
                const int orow = px / FR_OC, ocol = px - orow * FR_OC;
                const int gy = ty0 + orow, gx = tx0 + ocol;
                if (px < FR_OR * FR_OC && gy < p.H && gx < p.W) {
                    const char* src = ybuf + ((orow + 1) * FR_HC + ocol + 1) * FR_PXB + 16 * part;
                    const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
                    store16(reinterpret_cast<T*>(p.out) + ((long)gy * p.W + gx) * p.ldo + 32 * j + 8 * part, make_uint4(lo.x, lo.y, hi.x, hi.y));
                }
            }
#endif
#endif
            FW_PH(7);
#pragma unroll
            for (int jj = 0; jj < NCH; ++jj)
                if (jj == j) cs[jj] += cj;    // static register indices: cs[j] with a run-time j would live in scratch
        }
    }

#ifdef FW_FRONT_STAMP
    if ((blockIdx.x == 3 || blockIdx.x == 131) && lane == 0 && (wave == 0 || wave == 5))
        printf("front c=%d wg %d wave %d tiles %ld total %llu | A %llu gemm %llu B1 %llu ywrite %llu B2 %llu dw %llu B3 %llu store %llu\n", CIN, (int)blockIdx.x,
               wave, (long)(t_hi - t_lo), __builtin_amdgcn_s_memtime() - t_begin, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]);
#endif
    // ---- SCA pooling: fixed-order wave reduction of the lanes' sums -> partial[workgroup][c] ------------------------------------
    if (p.partial) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            f32x4 v = cs[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += __shfl_xor(v[i], o);
            if (lane == 0) *reinterpret_cast<f32x4*>(p.partial + (long)blockIdx.x * C + 32 * j + 4 * wave) = v;
        }
    }
}

static int front_cus() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        return v;
    }();
    return n;
}

int pw_dw_gate_blocks(int H, int W) {
    const long tiles = (long)((H + FR_OR - 1) / FR_OR) * ((W + FR_OC - 1) / FR_OC);
    return (int)(tiles < front_cus() ? tiles : front_cus());
}

bool pw_dw_gate_eligible(int cin) { return cin == 64 || cin == 128; }

void launch_pw_dw_gate(DType dt, const PwDwParams& p, hipStream_t st) {
    if (!pw_dw_gate_eligible(p.cin) || p.H <= 0 || p.W <= 0 || (p.ldx % 4) || (p.ldo % 4) || !p.x || !p.wpk || !p.bias || !p.wdw_t || !p.bdw || !p.out)
        throw Error(1, "pw_dw_gate: shape not eligible");
    dim3 grid((unsigned)pw_dw_gate_blocks(p.H, p.W)), block(512);
#define FW_F(CIN)                                                                                     \
    do {                                                                                              \
        if (dt == DT_BF16)                                                                            \
            hipLaunchKernelGGL((pw_dw_gate_kernel<__bf16, CIN>), grid, block, 0, st, p);              \
        else                                                                                          \
            hipLaunchKernelGGL((pw_dw_gate_kernel<_Float16, CIN>), grid, block, 0, st, p);            \
    } while (0)
    if (p.cin == 64) FW_F(64);
    else FW_F(128);
#undef FW_F
    FW_HIP_CHECK(hipGetLastError());
}

// Host-side packer: conv1 weights w[2c][c] fp32 with the LayerNorm's affine part folded in (w' = w * ln_w[k]) ->
// [64-channel chunk j][32-channel K chunk][16-row tile (4)][lane][8], the A operand of v_mfma_f32_16x16x32.  Chunk j holds x1
// channels 32j..32j+31 (tiles 0, 1) and the x2 channels c + 32j.. they are gated with (tiles 2, 3).  K order inside a
// fragment: element e of lane l is channel 32 kc + 16 (e >> 2) + 4 (l >> 4) + (e & 3) - the order in which the kernel's
// 16-byte loads of the fp32 stream land in its B fragments.  bias_out[2c] = bias + w ln_b.
size_t pack_pw_dw_gate_weights(DType dt, const float* w, const float* bias, const float* ln_w, const float* ln_b, int c, uint16_t* dst,
                               float* bias_out) {
    const int nch = 2 * c / 64, kcs = c / 32;
    const size_t n = (size_t)nch * kcs * 4 * 64 * 8;
    if (!dst) return n;
    size_t o = 0;
    for (int j = 0; j < nch; ++j)
        for (int kc = 0; kc < kcs; ++kc)
            for (int ct = 0; ct < 4; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int cc = 16 * ct + (lane & 15);
                        const int co = cc < 32 ? 32 * j + cc : c + 32 * j + (cc - 32);
                        const int k = 32 * kc + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
                        dst[o++] = f32_to_operand(dt, w[(size_t)co * c + k] * ln_w[k]);
                    }
    for (int co = 0; co < 2 * c; ++co) {
        double a = bias[co];
        for (int k = 0; k < c; ++k) a += (double)w[(size_t)co * c + k] * ln_b[k];
        bias_out[co] = (float)a;
    }
    return n;
}

}  // namespace fw
