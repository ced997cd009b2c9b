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
#include "fw_internal.h"
#include "conv_common.h"

namespace fw {

constexpr int FR_HR = 16, FR_HC = 32;                 // halo tile
constexpr int FR_OR = 14, FR_OC = 30;                 // output tile
constexpr int FR_PXB = 136;                           // LDS bytes per pixel record: 64 channels + 8 pad
constexpr int FR_Y_BYTES = FR_HR * FR_HC * FR_PXB;    // 69632
constexpr int FR_STRIP = 7;                           // output rows per depthwise item
constexpr int FR_TAIL = 11 * 64 * 4;                  // per chunk: depthwise taps [9][64], depthwise bias [64], conv bias [64] (fp32)

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
                    unsigned w4[4];
                    int orow = (gq * 8) / FR_OC, ocol = gq * 8 - orow * FR_OC;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const bool ok = gq * 8 + e < FR_OR * FR_OC && ty0 + orow < p.H && tx0 + ocol < p.W;     // wave-uniform
                        unsigned u = 0;
                        if (ok) u = *reinterpret_cast<const unsigned short*>(ych + (orow * FR_HC + ocol) * FR_PXB);
                        if (e & 1) w4[e >> 1] |= u << 16; else w4[e >> 1] = u;
                        if (++ocol == FR_OC) {
                            ocol = 0;
                            ++orow;
                        }
                    }
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
#define FW_F(CG, MODE)                                                                            \
    do {                                                                                          \
        if (dt == DT_BF16)                                                                        \
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
        }
    }
    return bytes;
}

}  // namespace fw
