// LayerNorm -> 1x1 convolution -> depthwise 3x3 [-> gate] in one kernel:
//   * the first half of a NAFBlock (norm1, conv1 c -> 2c, conv2 depthwise, SimpleGate x1 * x2, pooled sums for the SCA) at the
//     64- and 128-channel levels (reference: the NAFNet the TAP driver loads, tap_denoise.py:299-333; oracle/nafnet_ref.py);
//   * Restormer's two depthwise stages (tap_denoise.py:299-333 loads it as the default TAP model; oracle/restormer_ref.py):
//     norm1 -> qkv 1x1 -> qkv_dwconv (MDTA), and norm2 -> project_in -> dwconv -> gelu(x1) * x2 (GDFN), at 48 and 96 channels.
//
// Unfused, the wide tensor between the 1x1 and the depthwise convolution is written and read once each (the largest tensor of
// either block: 2c, 3c or 5.3c channels) and the depthwise kernels gathered it at 2 TB/s.  Here it only exists in LDS:
//
//   * a persistent 512-thread workgroup walks 14 x 30-pixel output tiles; the 1x1 conv is evaluated on the 16 x 32 halo tile
//     (x1.22 MACs - they are cheap: K = c);
//   * phase A: wave w loads the fp32 stream of halo rows 2w, 2w+1 straight into registers (lane = pixel l & 15, quarter
//     l >> 4 of every 16-channel group: each load instruction covers 64 contiguous bytes per pixel), LayerNorm statistics
//     with two xor-shuffles, (x - mean) * rstd becomes the B fragments of v_mfma_f32_16x16x32 (c / 2 registers); the affine
//     part of the LayerNorm is folded into the 1x1 conv by the host (W' = W diag(gamma), b' = b + W beta: pack_pw_dw_blocks);
//   * the 1x1 conv's output channels are processed in chunks of 64 (gate modes: 32 x1 channels and the 32 x2 channels they are
//     gated with).  A chunk's parameters - weight fragments, conv bias, depthwise taps and bias - are one contiguous block
//     that arrives by LDS-DMA while the previous chunk is in its depthwise phase (two buffers);
//   * per chunk: the GEMM into 16 accumulator tiles, + bias, zero outside the image (the depthwise conv pads the 1x1 conv's
//     OUTPUT), typed, into a [512 px][136 B] LDS image (stride 34 dwords: 16 lanes of consecutive pixels cover all 32 banks);
//   * depthwise 3x3 from LDS: wave w owns channels 4w..4w+3 of each half of the chunk, so its filter taps are wave-uniform;
//     lane = (7-row strip, column); 27 ds_read_b64 per half feed 63 x 2 v_pk_fma_f32, one input row ahead of the FMAs;
//   * the result overwrites the wave's own slots of the LDS image and leaves in 16-byte pieces, 64 (gate) or 128 contiguous
//     bytes per pixel; the SimpleGate sums stay in registers for the SCA pooling (wave reduction at the end of the kernel ->
//     partial[workgroup][c], the fixed-order scheme of dwconv3x3_gate_kernel).
#include <cstdlib>
#include <type_traits>
#include "fw_internal.h"
#include "conv_common.h"

namespace fw {

constexpr int FR_HR = 16, FR_HC = 32;                 // halo tile
constexpr int FR_OR = 14, FR_OC = 30;                 // output tile
constexpr int FR_PXB = 136;                           // LDS bytes per pixel record: 64 channels + 8 pad
constexpr int FR_Y_BYTES = FR_HR * FR_HC * FR_PXB;    // 69632
constexpr int FR_STRIP = 7;                           // output rows per depthwise item
constexpr int FR_TAIL32 = 11 * 64 * 4;                // per chunk: depthwise taps [9][64], depthwise bias [64], conv bias [64] (fp32)
constexpr int FR_TAPS16 = 64 * 3 * 2 * 4;             // ... and the taps once more for the matrix-core depthwise phase: [64][3 tap rows] x {w1 << 16 | w0, w2}, f16 bits
constexpr int FR_TAIL = FR_TAIL32 + FR_TAPS16;

constexpr int pw_dw_block_bytes(int kc) { return (kc * 4096 + FR_TAIL + 1023) / 1024 * 1024; }


// CG: 16-channel groups of the input (c = 16 CG); MODE: PWDW_NONE / PWDW_GATE_MUL / PWDW_GATE_GELU
template <typename T, int CG, int MODE>
__global__ __launch_bounds__(512, 2) void pw_dw_kernel(const PwDwParams p) {
    constexpr int KC = (CG + 1) / 2;                  // 32-channel chunks of the contraction (the last one half empty when CG is odd)
    constexpr int PB = pw_dw_block_bytes(KC);         // bytes of a chunk's parameter block
    constexpr int NPIECE = PB / 1024;
    constexpr bool GATE = MODE != PWDW_NONE;
    constexpr int OUT_B = GATE ? 64 : 128;            // output bytes per pixel and chunk
    __shared__ __attribute__((aligned(16))) char ybuf[FR_Y_BYTES];
    __shared__ __attribute__((aligned(16))) char pbuf[2 * PB];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, q = lane & 15, sl = lane >> 4;
    const unsigned pb_lds = (unsigned)(size_t)(lds_ptr_t)pbuf;
    const char* blocks = reinterpret_cast<const char*>(p.blocks);
    const int nch = p.n_chunks;

    auto fetch = [&](int chunk, int buf) {            // this wave's pieces of a parameter block
        for (int i = wave; i < NPIECE; i += 8)
            glds16(blocks + (size_t)chunk * PB + i * 1024, lane * 16, pb_lds + buf * PB + i * 1024);
    };

    const int tiles_x = (p.W + FR_OC - 1) / FR_OC, tiles_y = (p.H + FR_OR - 1) / FR_OR;
    const long ntiles = (long)tiles_x * tiles_y;
    const long t_lo = blockIdx.x * ntiles / gridDim.x, t_hi = (long)(blockIdx.x + 1) * ntiles / gridDim.x;
    if (t_lo >= t_hi) return;
    fetch(0, 0);

    // depthwise item of this lane: strip (0/1) and column; lanes 60-63 idle
    const bool dw_on = lane < 2 * FR_OC;
    const int strip = lane >= FR_OC ? 1 : 0;
    const int col = dw_on ? lane - strip * FR_OC : 0;
    char* yrd = ybuf + ((FR_STRIP * strip) * FR_HC + col) * FR_PXB + 8 * wave;

    f32x4 cs[4];                                      // PWDW_GATE_MUL: pooled sums of up to four chunks (c <= 128)
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float inv_c = 1.0f / (float)(16 * CG);
    unsigned g = 0;                                   // chunks done: parameter buffer = g & 1

#ifdef FW_FRONT_STAMP   // diagnostic build: cycles per phase of one wave, printed at the end
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = __builtin_amdgcn_s_memtime(), t_begin = tq;
#define FW_PH(i)                                                     \
    do {                                                             \
        const unsigned long long now = __builtin_amdgcn_s_memtime(); \
        ph[i] += now - tq;                                           \
        tq = now;                                                    \
    } while (0)
#else
#define FW_PH(i)
#endif

    for (long tile = t_lo; tile < t_hi; ++tile) {
        const int tyi = (int)(tile / tiles_x);
        const int ty0 = tyi * FR_OR, tx0 = (int)(tile - (long)tyi * tiles_x) * FR_OC;

        // ---- phase A: this wave's 64 halo pixels -> normalised B fragments ------------------------------------------------
        uint4 xb[4][KC];
        unsigned inside = 0;
        constexpr int TB = CG > 4 ? 2 : 4;            // pixel tiles whose raw fp32 is in flight together (<= 64 registers)
#pragma unroll
        for (int t0 = 0; t0 < 4; t0 += TB) {
            f32x4 v[TB][CG];
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
                for (int m = 0; m < CG; ++m) v[tt][m] = *reinterpret_cast<const f32x4*>(src + 16 * m);
            }
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) {
                const int t = t0 + tt;
                float s = 0.f;
#pragma unroll
                for (int m = 0; m < CG; ++m) s += (v[tt][m][0] + v[tt][m][1]) + (v[tt][m][2] + v[tt][m][3]);
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                const float mean = s * inv_c;
                float ss = 0.f;
#pragma unroll
                for (int m = 0; m < CG; ++m) {
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
                        if (2 * kc + u < CG) {
                            const f32x4 n = v[tt][2 * kc + u] * rstd;
                            h[u] = Op<T>::pack4(n[0], n[1], n[2], n[3]);
                        } else {
                            h[u] = make_uint2(0u, 0u);       // K padding of an odd group count: zero activations on zero weights
                        }
                    }
                    xb[t][kc] = make_uint4(h[0].x, h[0].y, h[1].x, h[1].y);
                }
            }
        }
        FW_PH(0);

#pragma unroll 1
        for (int j = 0; j < nch; ++j, ++g) {
            const char* pb = pbuf + (g & 1) * PB;
            const float* tail = reinterpret_cast<const float*>(pb + KC * 4096);   // [9][64] taps, [64] depthwise bias, [64] conv bias
            // the chunk's parameters have landed (each wave waits for its own DMAs, then the barrier); the same barrier says the
            // previous chunk's store phase is done with ybuf
            FW_WAIT_VMCNT(0);
            __syncthreads();
            FW_PH(2);
            {   // next chunk's parameters (cyclic over the tiles of this workgroup)
                const bool last = j + 1 == nch;
                if (!(last && tile + 1 == t_hi)) fetch(last ? 0 : j + 1, (int)((g + 1) & 1));
            }
            // ---- 1x1 conv, chunk j: 64 output channels x this wave's 64 pixels -----------------------------------------------
            f32x4 acc[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            const uint4* wl = reinterpret_cast<const uint4*>(pb) + lane;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const uint4 wf = wl[(kc * 4 + ct) * 64];
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t][ct] = Op<T>::mfma16(wf, xb[t][kc], acc[t][ct]);
                }
            FW_PH(1);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int cc = 16 * ct + 4 * sl;                                   // channel of the chunk
                const f32x4 bs = *reinterpret_cast<const f32x4*>(tail + 10 * 64 + cc);
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

            // ---- depthwise 3x3 [+ gate]: chunk channels 4 wave .. + 3 (half 0) and 32 + the same (half 1) -------------------
            f32x2 x1a[FR_STRIP], x1b[FR_STRIP];
            f32x4 cj = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
#pragma clang fp contract(fast)
                const int ch0 = 32 * half + 4 * wave;
                f32x2 wa[9], wb[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(tail + k * 64 + ch0);
                    wa[k] = f32x2{wv[0], wv[1]};
                    wb[k] = f32x2{wv[2], wv[3]};
                }
                const f32x4 bv = *reinterpret_cast<const f32x4*>(tail + 9 * 64 + ch0);
                const f32x2 ba = {bv[0], bv[1]}, bb = {bv[2], bv[3]};
                f32x2 aa[FR_STRIP], ab[FR_STRIP];
#pragma unroll
                for (int o = 0; o < FR_STRIP; ++o) {
                    aa[o] = ba;
                    ab[o] = bb;
                }
                // one input row ahead of the FMAs; the scheduling barriers keep hipcc from hoisting all 27 reads (54 registers)
                char* yh = yrd + 64 * half;
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
                // The result overwrites the wave's own slot of the LDS image (this wave's channels: nobody else reads them, and
                // the wave's own reads are all issued) and leaves in 16-byte pieces below: 8-byte stores of eight waves straight
                // from registers were 16 partial writes per 128-byte line.
                if (GATE && half == 0) {
#pragma unroll
                    for (int o = 0; o < FR_STRIP; ++o) {
                        if constexpr (MODE == PWDW_GATE_GELU) {
                            x1a[o] = gelu_erf2(aa[o]);
                            x1b[o] = gelu_erf2(ab[o]);
                        } else {
                            x1a[o] = aa[o];
                            x1b[o] = ab[o];
                        }
                    }
                } else {
                    const int gx = tx0 + col;
#pragma unroll
                    for (int o = 0; o < FR_STRIP; ++o) {
                        const int gy = ty0 + FR_STRIP * strip + o;
                        f32x2 ga = aa[o], gb = ab[o];
                        if constexpr (GATE) {
                            ga = x1a[o] * ga;
                            gb = x1b[o] * gb;
                        }
                        if (dw_on) {
                            *reinterpret_cast<uint2*>(yrd + (GATE ? 0 : 64 * half) + ((o + 1) * FR_HC + 1) * FR_PXB) = Op<T>::pack4(ga[0], ga[1], gb[0], gb[1]);
                            if (MODE == PWDW_GATE_MUL && gy < p.H && gx < p.W) cj += f32x4{ga[0], ga[1], gb[0], gb[1]};
                        }
                    }
                }
            }
            FW_PH(5);
            __syncthreads();
            FW_PH(6);
#ifndef FW_FRONT_ABL_STORE   // timing only: no stores (the sums keep the arithmetic alive)
            bool transposed = false;
            if constexpr (MODE == PWDW_NONE) transposed = p.qT && j < p.t_chunks;
            if (transposed) {
                // q / k for the Gram kernel: 16 bytes = 8 pixels of one channel, pixels in tile order, zeros where the tile has none
                T* dst = reinterpret_cast<T*>(p.qT) + 64 * 8 * j;
                constexpr int GROUPS = (FR_OR * FR_OC + 7) / 8 + 3;   // 56: 448 pixels, a multiple of 32
                // a wave handles one pixel group per round (lane = channel): which pixels, and whether they exist, is scalar work
                const char* ych = ybuf + FR_PXB * (FR_HC + 1) + 2 * lane;
#pragma unroll 1
                for (int gq = wave; gq < GROUPS; gq += 8) {
                    // eight 2-byte reads issued back to back, then masked: a branch per read (pixels the tile does not have) made hipcc wait
                    // out every read's latency in turn - 56 round trips per wave and chunk, a third of the qkv front (round 3)
                    unsigned u[8], keep[8];
                    int off[8];
                    int orow = (gq * 8) / FR_OC, ocol = gq * 8 - orow * FR_OC;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const bool ok = gq * 8 + e < FR_OR * FR_OC && ty0 + orow < p.H && tx0 + ocol < p.W;     // wave-uniform
                        off[e] = ok ? (orow * FR_HC + ocol) * FR_PXB : 0;
                        keep[e] = ok ? 0xffffu : 0u;
                        if (++ocol == FR_OC) {
                            ocol = 0;
                            ++orow;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) u[e] = *reinterpret_cast<const unsigned short*>(ych + off[e]);
                    __builtin_amdgcn_sched_barrier(0);
                    unsigned w4[4];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) w4[e >> 1] = (u[e] & keep[e]) | ((u[e + 1] & keep[e + 1]) << 16);
                    store16(dst + (((long)tile * GROUPS + gq) * p.t_ld + lane) * 8, make_uint4(w4[0], w4[1], w4[2], w4[3]));
                }
            } else {
                constexpr int PPX = OUT_B / 16;               // 16-byte pieces per pixel
                T* obase = reinterpret_cast<T*>(p.out) + (OUT_B / 2) * (j - (MODE == PWDW_NONE && p.qT ? p.t_chunks : 0));
#pragma unroll
                for (int k = 0; k < (FR_OR * FR_OC * PPX + 511) / 512; ++k) {
                    const int i = tid + 512 * k;              // (output pixel, piece)
                    const int px = i / PPX, part = i - px * PPX;
                    const int orow = px / FR_OC, ocol = px - orow * FR_OC;
                    const int gy = ty0 + orow, gx = tx0 + ocol;
                    if (px < FR_OR * FR_OC && gy < p.H && gx < p.W) {
                        const char* src = ybuf + ((orow + 1) * FR_HC + ocol + 1) * FR_PXB + 16 * part;
                        const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
                        store16(obase + ((long)gy * p.W + gx) * p.ldo + 8 * part, make_uint4(lo.x, lo.y, hi.x, hi.y));
                    }
                }
            }
#endif
            FW_PH(7);
            if constexpr (MODE == PWDW_GATE_MUL) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    if (jj == j) cs[jj] += cj;    // static register indices: cs[j] with a run-time j would live in scratch
            }
        }
    }

#ifdef FW_FRONT_STAMP
    if ((blockIdx.x == 3 || blockIdx.x == 131) && lane == 0 && (wave == 0 || wave == 5))
        printf("front cg=%d mode %d wg %d wave %d tiles %ld total %llu | A %llu gemm %llu B1 %llu ywrite %llu B2 %llu dw %llu B3 %llu store %llu\n", CG, MODE,
               (int)blockIdx.x, wave, (long)(t_hi - t_lo), __builtin_amdgcn_s_memtime() - t_begin, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]);
#endif
    // ---- SCA pooling: fixed-order wave reduction of the lanes' sums -> partial[workgroup][c] ------------------------------------
    if constexpr (MODE == PWDW_GATE_MUL) {
        if (p.partial) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j >= nch) break;
                f32x4 v = cs[j];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += __shfl_xor(v[i], o);
                if (lane == 0) *reinterpret_cast<f32x4*>(p.partial + (long)blockIdx.x * (32 * nch) + 32 * j + 4 * wave) = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same kernel with the depthwise 3x3 on the MATRIX CORES: opt-in (FW_PW_DW_MFMA=1), correct and tested, NOT faster - measured below.
// In pw_dw_kernel the depthwise phase is VALU-issue-bound - a conversion per loaded value, half a packed FMA per MAC, ~1000 issue
// slots per 64-channel chunk and wave - and takes 55 % of the GDFN kernel (tools/front_stamps.py).  Per channel c and tap row dy,
//     D[m][n] += sum_k A_dy[m][k] * B[k][n],   m = output column (16), k = input column (32), n = image row (16)
//     A_dy[m][k] = w[c][dy][k - m] for 0 <= k - m <= 2, else 0     (a Toeplitz band built in registers with four v_perm_b32 from the
//                  taps - the byte selectors are per-lane constants)
//     B[k][n]    = y[c][n + dy][col0 + k]                          (one ds_read_b128 per lane from a CHANNEL-MAJOR LDS tile)
// is 3 v_mfma_f32_16x16x32 per channel and 16 x 16 outputs: 9 % of their MACs are useful and it is still half the VALU phase's
// cycles (tools/dw_mfma.py, profiles/r02_dw_mfma.json: 4.25k against 8.2k per 64 channels of a tile).  What the kernel needs for it:
//   * the 1x1 conv with its operands SWAPPED (A = the pixels' B fragments of phase A, B = the weight fragments - both keep their
//     k order): D holds four consecutive PIXELS of one channel per lane, i.e. 8-byte pieces of a channel-major row;
//   * a chunk in two halves of 32 channels (gate modes: the x1 half, then the x2 half), each through a [32 ch][16 + 2 rows][56]
//     f16 image (row stride 28 dwords: conflict-free b128 reads; the pad columns and rows are zeroed once and never written, so
//     the band's zero entries never meet a NaN); wave w owns channels 4w .. 4w + 3 of the half, as before;
//   * the gate on the D fragments (x1 of the first half stays in registers), the SCA sums likewise;
//   * a pixel-major way out: the wave's four channels of a pixel are 8 bytes of a [420 px][64 or 128 B (+ 8)] image from which the
//     workgroup stores 16-byte pieces as before.
// Taps are rounded to the operand type (like the 1x1 weights), products are exact in fp32, the sum is fp32.
// Measured (round 3, phase stamps of a Restormer 512 x 512 tile and A/B of whole forwards on one box, profiles/r03_ab/pw_dw_mfma.txt):
// like for like - the qkv front, no gate - the depthwise phase goes from 5.3k to 3.7k cycles per 64-channel chunk (1.6k with the
// output writes taken out); the 8.2k of the GDFN front that round 2's microbenchmark was set against also hold the erf GELU (2.8k)
// and the gate + output writes (2.0k), which stay.  The two extra barriers per chunk (a half's image must be read out before the
// next half overwrites it), the wait they add at the chunk's first barrier and the channel-major store give that back:
// Restormer tile 10.03 against 9.91 ms, NAFNet 1080p forward 14.15 against 14.08 ms.  A chunk of this kernel is ~16k cycles in eight
// phases of 0.6k - 2.8k each (phase A 1.6k, GEMM 2.0k, image store 1.9k, barriers 2.2k, depthwise 1.6k, GELU 2.8k, gate + output
// image 2.0k, global stores 0.6k): no single phase is worth more than a sixth.
constexpr int DM_RS = 56;                              // halves per image row
constexpr int DM_ROWS = FR_HR + 2;                     // 16 rows + the two the last output rows' taps reach into
constexpr int DM_PLANE = DM_ROWS * DM_RS * 2 + 16;     // bytes per channel: + 16 spreads the 16 channels of a GEMM store over the banks
constexpr int DM_Y_BYTES = 32 * DM_PLANE;              // 65024

template <typename T, int CG, int MODE>
__global__ __launch_bounds__(512, 2) void pw_dw_mfma_kernel(const PwDwParams p) {
    constexpr int KC = (CG + 1) / 2;
    constexpr int PB = pw_dw_block_bytes(KC);
    constexpr int NPIECE = PB / 1024;
    constexpr bool GATE = MODE != PWDW_NONE;
    constexpr int OUT_B = GATE ? 64 : 128;            // output bytes per pixel and chunk
    constexpr int OSTR = OUT_B + 8;                   // bytes per pixel record of the output image
    constexpr int O_BYTES = (FR_OR * FR_OC * OSTR + 15) / 16 * 16;
    __shared__ __attribute__((aligned(16))) char ych[DM_Y_BYTES];
    __shared__ __attribute__((aligned(16))) char obuf[O_BYTES];
    __shared__ __attribute__((aligned(16))) char pbuf[2 * PB];
    static_assert(DM_Y_BYTES + O_BYTES + 2 * PB <= 160 * 1024, "LDS");

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, q = lane & 15, sl = lane >> 4;
    const unsigned pb_lds = (unsigned)(size_t)(lds_ptr_t)pbuf;
    const char* blocks = reinterpret_cast<const char*>(p.blocks);
    const int nch = p.n_chunks;

    auto fetch = [&](int chunk, int buf) {
        for (int i = wave; i < NPIECE; i += 8)
            glds16(blocks + (size_t)chunk * PB + i * 1024, lane * 16, pb_lds + buf * PB + i * 1024);
    };

    const int tiles_x = (p.W + FR_OC - 1) / FR_OC, tiles_y = (p.H + FR_OR - 1) / FR_OR;
    const long ntiles = (long)tiles_x * tiles_y;
    const long t_lo = blockIdx.x * ntiles / gridDim.x, t_hi = (long)(blockIdx.x + 1) * ntiles / gridDim.x;
    if (t_lo >= t_hi) return;
    fetch(0, 0);
    // zero the image once: pad columns 32..55 and pad rows 16, 17 are never written again
    for (int i = tid; i < DM_Y_BYTES / 16; i += 512) reinterpret_cast<uint4*>(ych)[i] = make_uint4(0, 0, 0, 0);

    // band selectors of this lane's A fragment (row m = q, k = 8 sl + 2 i + hh): tap d = k - m in {0, 1, 2} or none
    unsigned sel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned v = 0;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int d = 8 * sl + 2 * i + hh - q;
            const unsigned pair = (d == 0) ? 0x0100u : (d == 1) ? 0x0302u : (d == 2) ? 0x0504u : 0x0706u;   // bytes of {S1 = (0, w2), S0 = (w1, w0)}
            v |= pair << (16 * hh);
        }
        sel[i] = v;
    }

    f32x4 cs[4];                                      // PWDW_GATE_MUL: pooled sums of up to four chunks (c <= 128), this wave's four channels
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float inv_c = 1.0f / (float)(16 * CG);
    unsigned g = 0;                                   // chunks done: parameter buffer = g & 1
#ifdef FW_FRONT_STAMP   // diagnostic build: cycles per phase of one wave, printed at the end (FW_PH: defined with pw_dw_kernel above)
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = __builtin_amdgcn_s_memtime(), t_begin = tq;
#endif

    for (long tile = t_lo; tile < t_hi; ++tile) {
        const int tyi = (int)(tile / tiles_x);
        const int ty0 = tyi * FR_OR, tx0 = (int)(tile - (long)tyi * tiles_x) * FR_OC;

        // ---- phase A: this wave's 64 halo pixels -> normalised fragments (pw_dw_kernel's, unchanged) ----------------------------
        uint4 xb[4][KC];
        constexpr int TB = CG > 4 ? 2 : 4;
#pragma unroll
        for (int t0 = 0; t0 < 4; t0 += TB) {
            f32x4 v[TB][CG];
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) {
                const int t = t0 + tt;
                const int gy = ty0 - 1 + 2 * wave + (t >> 1), gx = tx0 - 1 + 16 * (t & 1) + q;
                const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
                const float* src = p.x + ((long)cy * p.W + cx) * p.ldx + 4 * sl;
#pragma unroll
                for (int m = 0; m < CG; ++m) v[tt][m] = *reinterpret_cast<const f32x4*>(src + 16 * m);
            }
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) {
                const int t = t0 + tt;
                float s = 0.f;
#pragma unroll
                for (int m = 0; m < CG; ++m) s += (v[tt][m][0] + v[tt][m][1]) + (v[tt][m][2] + v[tt][m][3]);
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                const float mean = s * inv_c;
                float ss = 0.f;
#pragma unroll
                for (int m = 0; m < CG; ++m) {
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
                        if (2 * kc + u < CG) {
                            const f32x4 n = v[tt][2 * kc + u] * rstd;
                            h[u] = Op<T>::pack4(n[0], n[1], n[2], n[3]);
                        } else {
                            h[u] = make_uint2(0u, 0u);
                        }
                    }
                    xb[t][kc] = make_uint4(h[0].x, h[0].y, h[1].x, h[1].y);
                }
            }
        }
        FW_PH(0);
        // which of the lane's GEMM outputs lie inside the image (the depthwise conv pads the 1x1 conv's OUTPUT with zeros):
        // D fragment of pixel tile t = halo row 2 wave + (t >> 1), columns 16 (t & 1) + 4 sl + r
        unsigned inside = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int gy = ty0 - 1 + 2 * wave + (t >> 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gx = tx0 - 1 + 16 * (t & 1) + 4 * sl + r;
                inside |= (unsigned)(gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) << (4 * t + r);
            }
        }

#pragma unroll 1
        for (int j = 0; j < nch; ++j, ++g) {
            const char* pb = pbuf + (g & 1) * PB;
            const float* tail = reinterpret_cast<const float*>(pb + KC * 4096);            // [9][64] taps (unused here), [64] depthwise bias, [64] conv bias
            const uint2* taps16 = reinterpret_cast<const uint2*>(pb + KC * 4096 + FR_TAIL32);   // [64][3]
            FW_WAIT_VMCNT(0);
            __syncthreads();     // the chunk's parameters have landed; the previous chunk's store phase is done with obuf, its depthwise phase with ych
            {
                const bool last = j + 1 == nch;
                if (!(last && tile + 1 == t_hi)) fetch(last ? 0 : j + 1, (int)((g + 1) & 1));
            }
            FW_PH(2);
            const uint4* wl = reinterpret_cast<const uint4*>(pb) + lane;
            f32x4 x1[4][2];      // gate modes: the first half's depthwise outputs (GELU'd for the GDFN gate)
            f32x4 cj = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                // ---- 1x1 conv, channels 32 half .. + 31 of the chunk x this wave's 64 pixels, operands swapped: lane = (channel q of a
                //      16-channel tile, pixels 4 sl .. 4 sl + 3 of a 16-pixel tile) -------------------------------------------------------
                f32x4 acc[4][2];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) acc[t][c2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        const uint4 wf = wl[(kc * 4 + 2 * half + c2) * 64];
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t][c2] = Op<T>::mfma16(xb[t][kc], wf, acc[t][c2]);
                    }
                FW_PH(1);
                if (half) __syncthreads();   // every wave is done reading the first half's image
                FW_PH(4);
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const float bs = tail[10 * 64 + 32 * half + 16 * c2 + q];
                    char* yc = ych + (16 * c2 + q) * DM_PLANE + 8 * sl;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        f32x4 y = acc[t][c2] + bs;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (!((inside >> (4 * t + r)) & 1u)) y[r] = 0.f;
                        *reinterpret_cast<uint2*>(yc + ((2 * wave + (t >> 1)) * DM_RS + 16 * (t & 1)) * 2) = Op<T>::pack4(y[0], y[1], y[2], y[3]);
                    }
                }
                FW_PH(3);
                __syncthreads();
                FW_PH(4);

                // ---- depthwise 3x3 of channels 4 wave .. + 3 of the half: 3 tap rows x 2 column halves MFMAs per channel -----------------
                f32x4 d[4][2];
                {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(tail + 9 * 64 + 32 * half + 4 * wave);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) d[cc][0] = d[cc][1] = f32x4{bv[cc], bv[cc], bv[cc], bv[cc]};
                }
                uint2 tp[4][3];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) tp[cc][dy] = taps16[(32 * half + 4 * wave + cc) * 3 + dy];
                const char* yr = ych + (4 * wave) * DM_PLANE + (q * DM_RS + 8 * sl) * 2;
                // One ds_read_b128 per channel and column half: lane n holds row n, and the fragments of tap rows 1 and 2 are the same
                // registers shifted by one / two lanes within the 16-lane row (v_mov_b32 row_shl: lane n takes lane n + dy; lanes past the
                // row get zero - they belong to output rows 14, 15, which the 14-row tile does not have).  Reading row n + dy from LDS
                // instead made the phase LDS-bound: 48 KiB-reads per wave and chunk, ~4.0k cycles per chunk against the VALU phase's 5.3k
                // (without the gate; profiles/r03_ab/pw_dw_mfma.txt).
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    uint4 A[3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const uint2 t2 = tp[cc][dy];
                        A[dy] = make_uint4(__builtin_amdgcn_perm(t2.y, t2.x, sel[0]), __builtin_amdgcn_perm(t2.y, t2.x, sel[1]),
                                           __builtin_amdgcn_perm(t2.y, t2.x, sel[2]), __builtin_amdgcn_perm(t2.y, t2.x, sel[3]));
                    }
#pragma unroll
                    for (int xh = 0; xh < 2; ++xh) {
                        const uint4 B0 = *reinterpret_cast<const uint4*>(yr + cc * DM_PLANE + (16 * xh) * 2);
                        auto shl = [](unsigned v, auto dyc) {
                            constexpr int DY = decltype(dyc)::value;
                            return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + DY, 0xf, 0xf, true);
                        };
                        const uint4 B1 = make_uint4(shl(B0.x, std::integral_constant<int, 1>{}), shl(B0.y, std::integral_constant<int, 1>{}),
                                                    shl(B0.z, std::integral_constant<int, 1>{}), shl(B0.w, std::integral_constant<int, 1>{}));
                        const uint4 B2 = make_uint4(shl(B0.x, std::integral_constant<int, 2>{}), shl(B0.y, std::integral_constant<int, 2>{}),
                                                    shl(B0.z, std::integral_constant<int, 2>{}), shl(B0.w, std::integral_constant<int, 2>{}));
                        d[cc][xh] = Op<T>::mfma16(A[0], B0, d[cc][xh]);
                        d[cc][xh] = Op<T>::mfma16(A[1], B1, d[cc][xh]);
                        d[cc][xh] = Op<T>::mfma16(A[2], B2, d[cc][xh]);
                    }
                }
                // d[cc][xh][r] = output (row q, column 16 xh + 4 sl + r) of channel 4 wave + cc of the half
                if (GATE && half == 0) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                        for (int xh = 0; xh < 2; ++xh) {
                            if constexpr (MODE == PWDW_GATE_GELU) {
                                const f32x2 a = gelu_erf2(f32x2{d[cc][xh][0], d[cc][xh][1]}), b = gelu_erf2(f32x2{d[cc][xh][2], d[cc][xh][3]});
                                x1[cc][xh] = f32x4{a[0], a[1], b[0], b[1]};
                            } else {
                                x1[cc][xh] = d[cc][xh];
                            }
                        }
                } else {
#pragma unroll
                    for (int xh = 0; xh < 2; ++xh)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ocol = 16 * xh + 4 * sl + r;
                            f32x4 o;
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) o[cc] = GATE ? x1[cc][xh][r] * d[cc][xh][r] : d[cc][xh][r];
                            if (q < FR_OR && ocol < FR_OC) {
                                *reinterpret_cast<uint2*>(obuf + (q * FR_OC + ocol) * OSTR + (GATE ? 0 : 64 * half) + 8 * wave) = Op<T>::pack4(o[0], o[1], o[2], o[3]);
                                if (MODE == PWDW_GATE_MUL && ty0 + q < p.H && tx0 + ocol < p.W) cj += o;
                            }
                        }
                }
            }
            FW_PH(5);
            __syncthreads();
            FW_PH(6);
            // ---- the chunk leaves in 16-byte pieces ---------------------------------------------------------------------------------------
            bool transposed = false;
            if constexpr (MODE == PWDW_NONE) transposed = p.qT && j < p.t_chunks;
            if (transposed) {
                // q / k for the Gram kernel: 16 bytes = 8 pixels of one channel, pixels in tile order, zeros where the tile has none
                T* dst = reinterpret_cast<T*>(p.qT) + 64 * 8 * j;
                constexpr int GROUPS = (FR_OR * FR_OC + 7) / 8 + 3;   // 56: 448 pixels, a multiple of 32
                const char* och = obuf + 2 * lane;
#pragma unroll 1
                for (int gq = wave; gq < GROUPS; gq += 8) {
                    unsigned u[8], keep[8];
                    int off[8];
                    int orow = (gq * 8) / FR_OC, ocol = gq * 8 - orow * FR_OC;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const bool ok = gq * 8 + e < FR_OR * FR_OC && ty0 + orow < p.H && tx0 + ocol < p.W;     // wave-uniform
                        off[e] = ok ? (orow * FR_OC + ocol) * OSTR : 0;
                        keep[e] = ok ? 0xffffu : 0u;
                        if (++ocol == FR_OC) {
                            ocol = 0;
                            ++orow;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) u[e] = *reinterpret_cast<const unsigned short*>(och + off[e]);
                    __builtin_amdgcn_sched_barrier(0);
                    unsigned w4[4];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) w4[e >> 1] = (u[e] & keep[e]) | ((u[e + 1] & keep[e + 1]) << 16);
                    store16(dst + (((long)tile * GROUPS + gq) * p.t_ld + lane) * 8, make_uint4(w4[0], w4[1], w4[2], w4[3]));
                }
            } else {
                constexpr int PPX = OUT_B / 16;               // 16-byte pieces per pixel
                T* obase = reinterpret_cast<T*>(p.out) + (OUT_B / 2) * (j - (MODE == PWDW_NONE && p.qT ? p.t_chunks : 0));
#pragma unroll
                for (int k = 0; k < (FR_OR * FR_OC * PPX + 511) / 512; ++k) {
                    const int i = tid + 512 * k;              // (output pixel, piece)
                    const int px = i / PPX, part = i - px * PPX;
                    const int orow = px / FR_OC, ocol = px - orow * FR_OC;
                    const int gy = ty0 + orow, gx = tx0 + ocol;
                    if (px < FR_OR * FR_OC && gy < p.H && gx < p.W) {
                        const char* src = obuf + px * OSTR + 16 * part;
                        const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
                        store16(obase + ((long)gy * p.W + gx) * p.ldo + 8 * part, make_uint4(lo.x, lo.y, hi.x, hi.y));
                    }
                }
            }
            FW_PH(7);
            if constexpr (MODE == PWDW_GATE_MUL) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    if (jj == j) cs[jj] += cj;
            }
        }
    }
#ifdef FW_FRONT_STAMP
    if ((blockIdx.x == 3 || blockIdx.x == 131) && lane == 0 && (wave == 0 || wave == 5))
        printf("front-mfma cg=%d mode %d wg %d wave %d tiles %ld total %llu | A %llu gemm %llu B1 %llu ywrite %llu B2 %llu dw+out %llu B3 %llu store %llu\n", CG, MODE,
               (int)blockIdx.x, wave, (long)(t_hi - t_lo), __builtin_amdgcn_s_memtime() - t_begin, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]);
#endif

    // ---- SCA pooling: fixed-order wave reduction of the lanes' sums -> partial[workgroup][c] ------------------------------------
    if constexpr (MODE == PWDW_GATE_MUL) {
        if (p.partial) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j >= nch) break;
                f32x4 v = cs[j];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += __shfl_xor(v[i], o);
                if (lane == 0) *reinterpret_cast<f32x4*>(p.partial + (long)blockIdx.x * (32 * nch) + 32 * j + 4 * wave) = v;
            }
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

long pw_dw_transposed_pixels(int H, int W) {
    return (long)((H + FR_OR - 1) / FR_OR) * ((W + FR_OC - 1) / FR_OC) * 448;
}

int pw_dw_blocks(int H, int W) {
    const long tiles = (long)((H + FR_OR - 1) / FR_OR) * ((W + FR_OC - 1) / FR_OC);
    return (int)(tiles < front_cus() ? tiles : front_cus());
}

bool pw_dw_eligible(int cin, int mode) {
    if (mode == PWDW_GATE_MUL) return cin == 64 || cin == 128;
    if (mode == PWDW_NONE || mode == PWDW_GATE_GELU) return cin == 48 || cin == 96;
    return false;
}

void launch_pw_dw(DType dt, const PwDwParams& p, hipStream_t st) {
    if (!pw_dw_eligible(p.cin, p.mode) || p.H <= 0 || p.W <= 0 || (p.ldx % 4) || (p.ldo % 8) || !p.x || !p.blocks || !p.out || p.n_chunks < 1 ||
        (p.mode == PWDW_GATE_MUL && p.n_chunks > 4) || (p.qT && (p.mode != PWDW_NONE || p.t_chunks < 1 || p.t_chunks >= p.n_chunks || p.t_ld < 64 * p.t_chunks)))
        throw Error(1, "pw_dw: shape not eligible");
    dim3 grid((unsigned)pw_dw_blocks(p.H, p.W)), block(512);
    // FW_PW_DW_MFMA=1: the depthwise phase on the matrix cores (pw_dw_mfma_kernel; not faster, see there); read per launch (tests flip
    // it inside one process)
    bool on_mfma = false;
    if (const char* e = getenv("FW_PW_DW_MFMA")) on_mfma = atoi(e) != 0;
#define FW_F(CG, MODE)                                                                            \
    do {                                                                                          \
        if (on_mfma) {                                                                            \
            if (dt == DT_BF16)                                                                    \
                hipLaunchKernelGGL((pw_dw_mfma_kernel<__bf16, CG, MODE>), grid, block, 0, st, p); \
            else                                                                                  \
                hipLaunchKernelGGL((pw_dw_mfma_kernel<_Float16, CG, MODE>), grid, block, 0, st, p); \
        } else if (dt == DT_BF16)                                                                 \
            hipLaunchKernelGGL((pw_dw_kernel<__bf16, CG, MODE>), grid, block, 0, st, p);          \
        else                                                                                      \
            hipLaunchKernelGGL((pw_dw_kernel<_Float16, CG, MODE>), grid, block, 0, st, p);        \
    } while (0)
    if (p.mode == PWDW_GATE_MUL) {
        if (p.cin == 64) FW_F(4, PWDW_GATE_MUL);
        else FW_F(8, PWDW_GATE_MUL);
    } else if (p.mode == PWDW_NONE) {
        if (p.cin == 48) FW_F(3, PWDW_NONE);
        else FW_F(6, PWDW_NONE);
    } else {
        if (p.cin == 48) FW_F(3, PWDW_GATE_GELU);
        else FW_F(6, PWDW_GATE_GELU);
    }
#undef FW_F
    FW_HIP_CHECK(hipGetLastError());
}

// Host-side packer.  w[N][c] fp32 (1x1 conv rows in the layout of the output tensor: N a multiple of 64; gate modes: x1 rows at
// 0, x2 rows at N / 2), bias[N] or null, LayerNorm weight / bias [c], depthwise filters wdw[N][9] and bias bdw[N] or null.
// Output: N / 64 parameter blocks of pw_dw_block_bytes((c / 16 + 1) / 2) bytes:
//   [K chunk][16-row tile (4)][lane][8] operand-typed A fragments of v_mfma_f32_16x16x32 with the LayerNorm weight folded in
//   (w' = w * ln_w[k]); K order inside a fragment: element e of lane l is channel 32 kc + 16 (e >> 2) + 4 (l >> 4) + (e & 3) - the
//   order in which the kernel's 16-byte loads of the fp32 stream land in its B fragments; channels >= c are zero;
//   then fp32 [9][64] depthwise taps, [64] depthwise bias, [64] conv bias (bias + w ln_b).
// Chunk j, chunk channel cc: gate -> row (cc < 32 ? 32 j + cc : N / 2 + 32 j + cc - 32); otherwise row 64 j + cc.
size_t pack_pw_dw_blocks(DType dt, const float* w, const float* bias, const float* ln_w, const float* ln_b, const float* wdw, const float* bdw,
                         int N, int c, int gate, void* dst_v) {
    const int kcs = (c / 16 + 1) / 2, nch = N / 64, pb = pw_dw_block_bytes(kcs);
    const size_t bytes = (size_t)nch * pb;
    if (!dst_v) return bytes;
    char* dst = static_cast<char*>(dst_v);
    memset(dst, 0, bytes);
    for (int j = 0; j < nch; ++j) {
        uint16_t* wf = reinterpret_cast<uint16_t*>(dst + (size_t)j * pb);
        float* tail = reinterpret_cast<float*>(dst + (size_t)j * pb + (size_t)kcs * 4096);
        auto row_of = [&](int cc) { return gate ? (cc < 32 ? 32 * j + cc : N / 2 + 32 * j + (cc - 32)) : 64 * j + cc; };
        size_t o = 0;
        for (int kc = 0; kc < kcs; ++kc)
            for (int ct = 0; ct < 4; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int row = row_of(16 * ct + (lane & 15));
                        const int k = 32 * kc + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
                        wf[o++] = f32_to_operand(dt, k < c ? w[(size_t)row * c + k] * ln_w[k] : 0.f);
                    }
        for (int cc = 0; cc < 64; ++cc) {
            const int row = row_of(cc);
            for (int k = 0; k < 9; ++k) tail[k * 64 + cc] = wdw[(size_t)row * 9 + k];
            tail[9 * 64 + cc] = bdw ? bdw[row] : 0.f;
            double a = bias ? bias[row] : 0.0;
            for (int k = 0; k < c; ++k) a += (double)w[(size_t)row * c + k] * ln_b[k];
            tail[10 * 64 + cc] = (float)a;
            // the taps in the operand type for the matrix-core depthwise phase: per tap row {w1 << 16 | w0, w2}
            uint32_t* t16 = reinterpret_cast<uint32_t*>(dst + (size_t)j * pb + (size_t)kcs * 4096 + FR_TAIL32) + cc * 6;
            for (int dy = 0; dy < 3; ++dy) {
                const uint32_t w0 = f32_to_operand(dt, wdw[(size_t)row * 9 + dy * 3]), w1 = f32_to_operand(dt, wdw[(size_t)row * 9 + dy * 3 + 1]),
                               w2 = f32_to_operand(dt, wdw[(size_t)row * 9 + dy * 3 + 2]);
                t16[2 * dy] = w0 | (w1 << 16);
                t16[2 * dy + 1] = w2;
            }
        }
    }
    return bytes;
}

}  // namespace fw
