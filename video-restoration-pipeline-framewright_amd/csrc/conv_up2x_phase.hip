// lrelu(conv3x3(nearest_x2(x))) - conv_up1 / conv_up2 of the RRDBNet tail (reference: src/framewright/processors/
// aesrgan_face.py:258-266, `F.interpolate(feat, scale_factor=2, mode="nearest")` followed by a 3x3 convolution) - evaluated on the
// LOW-resolution grid as four 2x2 "phase" convolutions.
//
// A nearest-x2 upsampled image repeats every source pixel 2x2 times, so of the nine taps of output pixel (2y + a, 2x + b) the
// three tap rows land on only two source rows and the three tap columns on only two source columns:
//   a = 0: rows 2y-1, 2y, 2y+1  -> source rows y-1 (w[0]), y (w[1] + w[2])        a = 1: 2y, 2y+1, 2y+2 -> y (w[0] + w[1]), y+1 (w[2])
// (columns alike).  The zero padding agrees: high-resolution row -1 / 2H is source row -1 / H.  Each phase (a, b) is a 2x2
// convolution with summed weights (summed in fp64 on the host, rounded to the operand type ONCE): 4 instead of 9 MFMA taps per
// output pixel, -2.25x on the two up-convs (4.1 % of the frame's MACs).  conv3x3_mfma.hip's `upsample2x` path gathers the
// repeated pixels through its DMA addresses and contracts all nine taps; it stays for the callers outside the RRDBNet.
//
// In terms of the 18 x 34 source halo tile (row 0 = source row y0 - 1) phase (a, b) of tile pixel (i, j) reads halo positions
// (i + a + r, j + b + s), r, s in {0, 1} - the 3x3-tap positions ty in {a, a+1}, tx in {b, b+1}.
//
// One workgroup (8 waves x 2 rows x 32 pixels, as the other conv kernels) owns a 16 x 32 SOURCE tile = a 32 x 64 output tile.
// Both 32-channel chunks of the source tile stay in LDS for the whole tile (read from HBM once; the gather form re-reads every
// source pixel four times); the tile is four items (a, chunk) of 128 MFMAs per wave, each holding the two phases b = 0, 1 of row
// phase a in 8 accumulator tiles (4 output-channel tiles x 2 phases):
//   tile k (even): (a0, c0) (a0, c1) emit a = 0 | (a1, c1) (a1, c0) emit a = 1        tile k + 1: chunk order swapped
// so that stage c holds chunk c, a stage's refill for the next tile goes out right after its last use and every DMA is issued
// one item ahead of its use (the chunk read last in a tile is read first in the next one).  Weights: 32 fragments (32 KiB) per
// item, double-buffered.  LDS: 2 x 40 KiB + 2 x 32 KiB = 144 KiB.
#include <cstdlib>
#include <vector>
#include "fw_internal.h"
#include "conv_common.h"

#ifndef FW_UP_SLOT_W
#define FW_UP_SLOT_W 2
#define FW_UP_SLOT_A 6
#endif
#ifndef FW_UP_FULL_LINES
#define FW_UP_FULL_LINES 1
#endif

namespace fw {

constexpr int UP_WFRAGS = 32;  // per item: 8 (tx, r, b) steps x 4 output-channel tiles
struct UpSmem {
    static constexpr int W_REGION = UP_WFRAGS * 64;
    static constexpr int W_BASE = 2 * ACT_REGION;
    static constexpr int TOTAL = 2 * ACT_REGION + 2 * W_REGION;  // 9216 pieces = 147456 bytes
};
static_assert(NWAVES == 8 && RPW == 2 && ACT_ITERS == 5, "written for 8 waves of 2 rows");
static_assert(UpSmem::TOTAL * 16 <= 160 * 1024, "LDS");

// step s of an item, in use order: halo column offset tx, row tap r, column phase b (column tap = tx - b)
__host__ __device__ constexpr int up_tx(int s) { return s < 2 ? 0 : (s < 6 ? 1 : 2); }
__host__ __device__ constexpr int up_r(int s) { return s < 2 ? s : (s < 6 ? (s - 2) >> 1 : s - 6); }
__host__ __device__ constexpr int up_b(int s) { return s < 2 ? 0 : (s < 6 ? (s - 2) & 1 : 1); }

// One item: chunk c (32 input channels) x row phase a: acc[row][4 b + w][ph] += W[step(tx, r, b)][w] * halo(row + r, px + tx).
// `a` = the stage + a * ROW_PIECES (row phase 1 reads one halo row lower), `wl` = the weight stage + lane (fragment k = 4 step + w).
// The B fragments of the next tx are read into the other half of xr while the current one is contracted.
template <typename T, typename Slot>
__device__ __forceinline__ void conv_item_up(f32x4 (&acc)[RPW][8][2], const uint4* a, const uint4* wl, const int (&rd_off)[3][2],
                                             Slot dma_slot) {
#ifndef FW_UP_WRING
#define FW_UP_WRING 3
#endif
    constexpr int RING = FW_UP_WRING;
    // Four register rows for the three halo rows of the current tx: the spare one takes row 1 of the next tx early, rows 0 and 2
    // follow into the slots that fall free (row 0 after the last r = 0 step, rows 1 / 2 after the last step of the tx).  Which slot
    // holds which row is static after unrolling: tx0 (0, 1, 2), tx1 (0, 3, 1), tx2 (0, 2, 3).
    uint4 xr[4][2];
    uint4 wf[RING];
    auto slot_of = [](int tx, int h) { return h == 0 ? 0 : (tx == 0 ? h : (tx == 1 ? (h == 1 ? 3 : 1) : (h == 1 ? 2 : 3))); };
    auto load_w = [&](int k) { wf[k % RING] = wl[k * 64]; };
    auto load_x = [&](int tx, int h) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) xr[slot_of(tx, h)][ph] = a[h * ROW_PIECES + rd_off[tx][ph]];
    };
#pragma unroll
    for (int h = 0; h < 3; ++h) load_x(0, h);
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) load_w(k);
    FW_SB();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int tx = up_tx(s), r = up_r(s), b = up_b(s);
        if (s == 0) load_x(1, 1);   // spare slot 3
        if (s == 1) load_x(1, 0);   // slot 0: free since step 0 (r = 0 reads rows 0, 1)
        if (s == 2) {
            load_x(1, 2);           // slot 1: free since step 1; read from step 4 on
            load_x(2, 1);           // spare slot 2
        }
        if (s == 4) load_x(2, 0);   // slot 0: free since step 3
        if (s == 6) load_x(2, 2);   // slot 3: free since step 5; read in step 7
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int k = s * 4 + w;
            if (k + RING - 1 < UP_WFRAGS) load_w(k + RING - 1);
            FW_SB();
#pragma unroll
            for (int row = 0; row < RPW; ++row)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph)
                    acc[row][4 * b + w][ph] = Op<T>::mfma16(wf[k % RING], xr[slot_of(tx, row + r)][ph], acc[row][4 * b + w][ph]);
            FW_SB();
            dma_slot(k);
            FW_SB();
        }
    }
}

template <typename T>
__global__ __launch_bounds__(64 * NWAVES, WAVES_PER_SIMD) void conv_up2x_phase_kernel(const ConvUpParams p) {
    using SM = UpSmem;
    __shared__ __attribute__((aligned(16))) uint4 lds[SM::TOTAL];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int q = lane & 15;
    const int sl = lane >> 4;

    const int NB = gridDim.x;
    const int xcd = blockIdx.x & 7;
    const int qn = NB >> 3, rn = NB & 7;
    const int lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int tiles_x = (p.W + TILE_W - 1) / TILE_W;
    const int tiles_y = (p.H + TILE_H - 1) / TILE_H;
    const int ntiles = tiles_x * tiles_y;
    const int t_lo = (int)((long)lb * ntiles / NB);
    const int t_hi = (int)((long)(lb + 1) * ntiles / NB);
    if (t_lo >= t_hi) return;
    const int nitems = (t_hi - t_lo) * 4;

    // ---- per-lane DMA plan of the 18 x 34 source halo (conv3x3_mfma.hip without the gather) ------------------------------------
    unsigned relb[ACT_ITERS];
#pragma unroll
    for (int i = 0; i < ACT_ITERS; ++i) {
        const int idx = (ACT_ITERS * wave + i) * 64 + lane;
        const int row = idx / ROW_PIECES;
        const int rm = idx - row * ROW_PIECES;
        const int px = rm >> 2;
        const int s = (rm & 3) ^ halo_swz(px);
        relb[i] = (unsigned)(((row * p.W + px) * p.in_cstride + s * 8) * 2) + (unsigned)(4 - i) * 1024u;
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
    const char* in = reinterpret_cast<const char*>(p.in);
    const char* w_b = reinterpret_cast<const char*>(p.wpk);
    const unsigned lane16 = lane * 16;
    const long chunk_bytes = p.in_pstride * 2;

    // chunk c of tile t into stage c
    auto issue_act = [&](int t, int c) {
        int ty, tx;
        tile_pos(t, tiles_x, tiles_y, &ty, &tx);
        const int ty0 = ty * TILE_H, tx0 = tx * TILE_W;
        const char* src = in + (long)c * chunk_bytes + ((long)(ty0 - 1) * p.W + (tx0 - 1)) * p.in_cstride * 2;
        unsigned ok = 0;
#pragma unroll
        for (int i = 0; i < ACT_ITERS; ++i) {
            const int idx = (ACT_ITERS * wave + i) * 64 + lane;
            const int row = idx / ROW_PIECES;
            const int px = (idx - row * ROW_PIECES) >> 2;
            if (idx < ACT_PIECES && (unsigned)(ty0 - 1 + row) < (unsigned)p.H && (unsigned)(tx0 - 1 + px) < (unsigned)p.W) ok |= 1u << i;
        }
        const bool all = __builtin_amdgcn_readfirstlane(__all(ok == (1u << ACT_ITERS) - 1u)) != 0;
        const unsigned dst = (unsigned)(c * ACT_REGION + ACT_ITERS * wave * 64);
        if (all) {
            glds16_batch_a(src, relb, lds_base + (dst + 4 * 64) * 16u);
        } else {
#pragma unroll
            for (int i = 0; i < ACT_ITERS; ++i)
                glds16_v(((ok >> i) & 1u) ? src + (relb[i] - (unsigned)(4 - i) * 1024u) : reinterpret_cast<const char*>(p.zeros),
                         lds_base + (dst + i * 64) * 16u);
        }
    };
    // the 32 fragments of item (a, c) into weight stage ws: four per wave, one batch
    auto issue_w = [&](int a, int c, int ws) {
        const int f4 = 4 * wave + 4;
        glds16_batch_w<4>(w_b + (size_t)((a * 2 + c) * UP_WFRAGS + f4) * 1024, lane16,
                          lds_base + (unsigned)(SM::W_BASE + ws * SM::W_REGION + f4 * 64) * 16u);
    };

    int rd_off[3][2];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int px = 16 * ph + q + tx;
            rd_off[tx][ph] = (RPW * wave) * ROW_PIECES + px * 4 + (sl ^ halo_swz(px));
        }

    f32x4 acc[RPW][8][2];  // [row][4 * column phase b + 16-channel tile w][16-pixel half]
    const int ls = (sl & 1) ? 2 + (sl >> 1) : (sl >> 1);   // the 8-channel slot this lane holds after the permlane swap
    const long Wout = 2L * p.W;
    char* outb = reinterpret_cast<char*>(p.out);
    [[maybe_unused]] const unsigned lane_off = (unsigned)((2 * q * p.out_cstride + ls * 8) * 2);
    const bool odd = (q & 1) != 0;
    const unsigned lane_off1 = (unsigned)(((odd ? 2 * q - 1 : 2 * q) * p.out_cstride + ls * 8) * 2);       // FW_UP_FULL_LINES: pixels 4k / 4k + 1
    const unsigned lane_off2 = (unsigned)(((odd ? 2 * q + 1 : 2 * q + 2) * p.out_cstride + ls * 8) * 2);   //                   pixels 4k + 2 / 4k + 3

    issue_w(0, 0, 0);
    issue_act(t_lo, 0);
    issue_act(t_lo, 1);

    int n = 0;
    for (int t = t_lo; t < t_hi; ++t) {
        const int A = (t - t_lo) & 1, B = A ^ 1;
        int tty, ttx;
        tile_pos(t, tiles_x, tiles_y, &tty, &ttx);
        const int y0 = tty * TILE_H, x0 = ttx * TILE_W;
#pragma clang loop unroll(disable)
        for (int j = 0; j < 4; ++j, ++n) {
            const int a = j >> 1;
            const int c = (j == 0 || j == 3) ? A : B;
            if ((j & 1) == 0) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + 16 * w + 4 * sl);
#pragma unroll
                    for (int row = 0; row < RPW; ++row)
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph) {
                            acc[row][w][ph] = bv;
                            acc[row][4 + w][ph] = bv;
                        }
                }
            }
            // the DMAs of the items that follow an emit were waited for ahead of its stores (vmcnt counts stores too)
            if ((j & 1) == 1 || (j == 0 && t == t_lo)) FW_WAIT_VMCNT(0);
            __syncthreads();
            const bool more = n + 1 < nitems;
            const int na_ = j < 3 ? (j + 1) >> 1 : 0;
            const int nc_ = (j == 2) ? A : B;   // j = 0, 1 -> B; j = 2 -> A; j = 3 -> the next tile's first chunk, this tile's B
            const bool act_next = j == 3 && t + 1 < t_hi;   // chunk B of the next tile (its first): stage B was last read in item 2
            const bool act_this = j == 0 && t > t_lo;       // chunk B of this tile: stage B = the previous tile's A, last read in its item 3
            // The activation refill goes out BEFORE the item's fragments are live (128 accumulator registers + 32 of B fragments +
            // the weight ring leave no room for its per-lane addresses: issued from a slot inside the item it spilled 26
            // registers); the weights' single batch goes out in the shadow of the first MFMAs.
            if (act_next) issue_act(t + 1, B);
            if (act_this) issue_act(t, B);
            auto dma_slot = [&](int d) {
                if (d == FW_UP_SLOT_W) {
                    if (more) issue_w(na_, nc_, (n + 1) & 1);
                }
            };
            const uint4* act = lds + c * ACT_REGION + a * ROW_PIECES;
            const uint4* wl = lds + SM::W_BASE + (n & 1) * SM::W_REGION + lane;
            conv_item_up<T>(acc, act, wl, rd_off, dma_slot);
            if (j & 1) {
                // row phase a is complete: the DMAs issued during this item first, then the stores
                FW_WAIT_VMCNT(0);
#pragma unroll
                for (int row = 0; row < RPW; ++row) {
                    const int y = y0 + RPW * wave + row;
                    if (y >= p.H) continue;  // wave-uniform
                    char* rowbase = outb + ((long)(2 * y + a) * Wout + 2 * x0) * p.out_cstride * 2;
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) {
                            uint4 v[2];
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                f32x4 oa = acc[row][4 * b + 2 * c2][ph], ob = acc[row][4 * b + 2 * c2 + 1][ph];
                                if (p.act) {
                                    oa = lrelu4(oa);
                                    ob = lrelu4(ob);
                                }
                                const uint2 pa = Op<T>::pack4(oa[0], oa[1], oa[2], oa[3]);
                                const uint2 pb = Op<T>::pack4(ob[0], ob[1], ob[2], ob[3]);
                                const u32x2 sx = __builtin_amdgcn_permlane16_swap(pa.x, pb.x, false, false);
                                const u32x2 sy = __builtin_amdgcn_permlane16_swap(pa.y, pb.y, false, false);
                                v[b] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                            }
                            char* base = rowbase + (long)c2 * p.out_pstride * 2 + (long)(32 * ph) * p.out_cstride * 2;
#if FW_UP_FULL_LINES
                            // A lane holds its slot of output pixels 2x (v[0]) and 2x + 1 (v[1]); stored as they stand, every store
                            // instruction writes 64-byte halves of 16 lines.  Neighbouring lanes x, x ^ 1 trade one of the two
                            // (quad-permute DPP, no LDS): the first store then covers pixels 4k, 4k + 1 and the second 4k + 2,
                            // 4k + 3 - whole 128-byte lines from 8 lanes each.
                            const uint4 snd = odd ? v[0] : v[1];
                            uint4 rcv;
                            rcv.x = (unsigned)__builtin_amdgcn_mov_dpp((int)snd.x, 0xB1, 0xF, 0xF, true);
                            rcv.y = (unsigned)__builtin_amdgcn_mov_dpp((int)snd.y, 0xB1, 0xF, 0xF, true);
                            rcv.z = (unsigned)__builtin_amdgcn_mov_dpp((int)snd.z, 0xB1, 0xF, 0xF, true);
                            rcv.w = (unsigned)__builtin_amdgcn_mov_dpp((int)snd.w, 0xB1, 0xF, 0xF, true);
                            const int xs = x0 + 16 * ph + q;                 // this lane's source pixel; the neighbour's is xs ^ 1
                            const bool own_ok = xs < p.W, nb_ok = (xs ^ 1) < p.W;
                            if (odd ? nb_ok : own_ok) store16(base + lane_off1, odd ? rcv : v[0]);
                            if (odd ? own_ok : nb_ok) store16(base + lane_off2, odd ? v[1] : rcv);
#else
                            if (x0 + 16 * ph + q < p.W) {
                                store16(base + lane_off, v[0]);
                                store16(base + (long)p.out_cstride * 2 + lane_off, v[1]);
                            }
#endif
                        }
                }
            }
        }
    }
}

int conv_num_cus();

void launch_conv_up2x_phase(DType dt, const ConvUpParams& p_in, hipStream_t stream) {
    ConvUpParams p = p_in;
    p.zeros = conv_zero_page();
    if (p.H <= 0 || p.W <= 0) throw Error(1, "conv_up2x_phase: empty problem");
    if (!p.in || !p.wpk || !p.bias || !p.out) throw Error(1, "conv_up2x_phase: NULL argument");
    if (p.in_cstride < 32 || (p.in_cstride & 7) || p.in_pstride < 32 || (p.in_pstride & 7) ||
        (p.in_pstride == 32 && p.in_cstride < 64 && (long)p.H * p.W > 1))
        throw Error(1, "conv_up2x_phase: bad input channel/plane stride");
    if (p.out_cstride < 32 || (p.out_cstride & 7) || p.out_pstride < 32 || (p.out_pstride & 7))
        throw Error(1, "conv_up2x_phase: bad output channel/plane stride");
    const int tiles = ((p.W + TILE_W - 1) / TILE_W) * ((p.H + TILE_H - 1) / TILE_H);
    const int cus = conv_num_cus();
    dim3 grid(tiles < cus ? tiles : cus), block(64 * NWAVES);
    if (dt == DT_BF16)
        hipLaunchKernelGGL((conv_up2x_phase_kernel<__bf16>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((conv_up2x_phase_kernel<_Float16>), grid, block, 0, stream, p);
    FW_HIP_CHECK(hipGetLastError());
}

// Fragment order: [row phase a][chunk c][step s = (tx, r, b) in use order][16-channel tile w][lane][j]; value = the sum of
// w[cout][cin][dy][dx] over the taps that phase (a, b) folds onto source offset (r, s_col = tx - b):
//   phase 0: tap 0 -> offset 0, taps 1, 2 -> offset 1;  phase 1: taps 0, 1 -> offset 0, tap 2 -> offset 1.
size_t pack_conv_up2x_phase_weights(DType dt, const float* w, uint16_t* dst) {
    const size_t n = (size_t)2 * 2 * UP_WFRAGS * 64 * 8;
    if (!dst) return n;
    auto taps = [](int phase, int off, int* lo, int* hi) {
        if (phase == 0) {
            *lo = off == 0 ? 0 : 1;
            *hi = off == 0 ? 0 : 2;
        } else {
            *lo = off == 0 ? 0 : 2;
            *hi = off == 0 ? 1 : 2;
        }
    };
    size_t o = 0;
    for (int a = 0; a < 2; ++a)
        for (int c = 0; c < 2; ++c)
            for (int s = 0; s < 8; ++s) {
                const int tx = up_tx(s), r = up_r(s), b = up_b(s);
                int y_lo, y_hi, x_lo, x_hi;
                taps(a, r, &y_lo, &y_hi);
                taps(b, tx - b, &x_lo, &x_hi);
                for (int wt = 0; wt < 4; ++wt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int co = 16 * wt + (lane & 15);
                            const int ci = 32 * c + 8 * (lane >> 4) + j;
                            double v = 0.0;
                            for (int dy = y_lo; dy <= y_hi; ++dy)
                                for (int dx = x_lo; dx <= x_hi; ++dx) v += (double)w[((size_t)co * 64 + ci) * 9 + dy * 3 + dx];
                            dst[o++] = f32_to_operand(dt, (float)v);
                        }
            }
    return n;
}

}  // namespace fw
