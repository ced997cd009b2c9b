// The fused conv pair (conv3x3_pair.hip: x_a = lrelu(conv_a([x..])), x_b = lrelu(conv_b([x.., x_a])), reference
// src/framewright/processors/aesrgan_face.py:184-187) as a SLIDING WINDOW down a column of tiles.
//
// conv3x3_pair.hip evaluates both convs on one 16x32 region and delivers its 14x30 interior: the one-pixel ring is
// recomputed by the neighbours (+22 % MFMA work).  Here the vertical part of that ring is gone: a workgroup walks DOWN a column
// with a step of 16 rows, conv_b runs ONE ROW BEHIND conv_a, and the two x_a rows a tile shares with the next one stay in LDS:
//
//   tile ty:   conv_a rows  A = [16 ty, 16 ty + 15]          -> x_a rows A
//              conv_b rows  B = [16 ty - 1, 16 ty + 14]      needs x_a rows [16 ty - 2, 16 ty + 15] = carry (2 rows) + A
//   input halo rows [16 ty - 2, 16 ty + 16] (19 rows): conv_a row r reads r-1..r+1, conv_b row r the same
//
// so a tile delivers 16 rows of x_a and 16 rows of x_b for 16 rows of conv_a and conv_b work (16/16 instead of 14/16; the
// horizontal ring stays: 30 of 32 columns).  Wave w owns conv_a rows 2w, 2w+1 and conv_b rows 2w-1, 2w of the tile: five
// halo rows (2w .. 2w+4) of B fragments instead of four, conv_a reading them one row lower than conv_b.
// The x_a tile keeps the standard 18-row halo format with row 0 = image row 16 ty - 2: rows 0,1 come from the carry buffer
// (wave 7 copies rows 16,17 of the tile there after the x_a item), rows 2..17 are this tile's conv_a.  A workgroup whose
// run starts in the middle of a column first runs the tile above it with conv_a only (no conv_b MFMAs, no x_a item, no
// stores) to obtain the carry rows; at the top of a column the carry is zero (= conv_b's zero padding).
#include <cstdlib>
#include <type_traits>
#include "fw_internal.h"
#include "conv_common.h"

#ifndef FW_DMA_SLOT_W
#define FW_DMA_SLOT_W 2
#define FW_DMA_SLOT_A 6
#endif

namespace fw {

constexpr int PS_TH = TILE_H;          // 16 rows per step
constexpr int PS_TW = TILE_W - 2;      // 30 valid pixels per tile row
constexpr int PS_HALO_H = TILE_H + 3;  // 19 input rows
constexpr int PS_PIECES = PS_HALO_H * ROW_PIECES;  // 2584
constexpr int PS_EXTRA = PS_PIECES - ACT_ITERS * NWAVES * 64;  // 24 pieces beyond the 40 batched KiB: one more DMA of wave 0
constexpr int PS_REGION = 41 * 64;     // pieces per activation stage
constexpr int PS_CARRY = 2 * ROW_PIECES;  // two x_a rows

struct SlideSmem {
    static constexpr int W_REGION = 2 * W_FRAGS * 64;
    static constexpr int W_BASE = 2 * PS_REGION;
    static constexpr int CARRY = W_BASE + 2 * W_REGION;
    static constexpr int TOTAL = CARRY + PS_CARRY;  // 10128 pieces = 162048 bytes
};
static_assert(NWAVES == 8 && RPW == 2 && PS_EXTRA > 0 && PS_EXTRA <= 64, "written for 8 waves of 2 rows");
static_assert(SlideSmem::TOTAL * 16 <= 160 * 1024, "LDS");

// One shared-chunk item with conv_b one row behind conv_a.  acc tiles 0,1 = conv_a rows (2w, 2w+1), 2,3 = conv_b rows
// (2w-1, 2w); xr[h] = halo row 2w + h.  conv_a row r, tap row dy reads xr[r + dy + 1], conv_b xr[r + dy].  Row h is last used
// at dy = min(h, 2): rows 0 / 1 are re-read for the next dx after dy = 0 / 1, row 2 after conv_b's tiles of dy = 2, rows 3, 4
// at the start of the next dx.  conv_b's tiles go first within a step so that the freshly read rows are needed last.
template <typename T, bool WITH_B, typename WIdx, typename Slot>
__device__ __forceinline__ void conv_item_lag(f32x4 (&acc)[RPW][4][2], const uint4* a, const uint4* wl, const int (&rd_off)[3][2],
                                              WIdx widx, Slot dma_slot) {
    constexpr int NWU = WITH_B ? 4 : 2;
    constexpr int NK = 9 * NWU;
    constexpr int RING = 3;
    uint4 xr[5][2];
    uint4 wf[RING];
    auto tile_of = [](int w) { return WITH_B ? ((w + 2) & 3) : w; };
    auto load_w = [&](int k) {
        const int t = k / NWU, w = k - t * NWU;
        const int dx = t / 3, dy = t - 3 * dx;
        wf[k % RING] = wl[widx(dy * 3 + dx, tile_of(w)) * 64];
    };
    auto load_row = [&](int h, int dx) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) xr[h][ph] = a[h * ROW_PIECES + rd_off[dx][ph]];
    };
#pragma unroll
    for (int h = 0; h < 5; ++h) load_row(h, 0);
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) load_w(k);
    FW_SB();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dx = t / 3, dy = t - 3 * dx;
        if (dx < 2 && dy == 1) load_row(0, dx + 1);
        if (dx < 2 && dy == 2) load_row(1, dx + 1);
        if (dx > 0 && dy == 0) {
            load_row(3, dx);
            load_row(4, dx);
        }
#pragma unroll
        for (int w = 0; w < NWU; ++w) {
            const int k = t * NWU + w;
            if (k + RING - 1 < NK) load_w(k + RING - 1);
            FW_SB();
            const int tile = tile_of(w);
            const int lag = tile >= 2 ? 0 : 1;
#pragma unroll
            for (int row = 0; row < RPW; ++row)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph)
                    acc[row][tile][ph] = Op<T>::mfma16(wf[k % RING], xr[row + dy + lag][ph], acc[row][tile][ph]);
            FW_SB();
#pragma unroll
            for (int d = 0; d < 4 / NWU; ++d) dma_slot((t * NWU + w) * (4 / NWU) + d);
            FW_SB();
            // row 2 is free once conv_b's tiles of dy = 2 are out (WITH_B: after w = 1; conv_a only: row 2 is read by conv_a
            // row 0 at dy = 1 and row 1 at dy = 0 only, so it is free at dy = 2 as well)
            if (dx < 2 && dy == 2 && w == (WITH_B ? 1 : 0)) load_row(2, dx + 1);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(64 * NWAVES, WAVES_PER_SIMD) void conv3x3_pair_slide_kernel(const ConvPairParams p) {
    using SM = SlideSmem;
    __shared__ __attribute__((aligned(16))) uint4 lds[SM::TOTAL];
    constexpr int NW = 4;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int q = lane & 15;
    const int sl = lane >> 4;
    FW_STAMP_INIT();  // phase stamps: diagnostic builds only (-DFW_PAIR_STAMP, conv_common.h)

    const int NB = gridDim.x;
    const int xcd = blockIdx.x & 7;
    const int qn = NB >> 3, rn = NB & 7;
    const int lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int tiles_x = (p.W + PS_TW - 1) / PS_TW;
    const int tiles_y = (p.H + PS_TH) / PS_TH;  // the last conv_b row H - 1 = 16 ty + 14 at the latest
    const int ntiles = tiles_x * tiles_y;        // column-major: t = tx * tiles_y + ty
    const int t_lo = (int)((long)lb * ntiles / NB);
    const int t_hi = (int)((long)(lb + 1) * ntiles / NB);
    if (t_lo >= t_hi) return;
#ifdef FW_PS_NOWARM   // timing only (wrong pixels in the first conv_b row of a run that starts mid-column): what the warm-up tiles cost
    const int needs_warm = 0;
#else
    const int needs_warm = (t_lo % tiles_y) != 0;  // the run starts below the top of a column: conv_a of the tile above first
#endif
    const int t_begin = t_lo - needs_warm;
    const int na = p.na;
    const int nitems = (t_hi - t_lo) * (na + 1) + needs_warm * na;

    // ---- per-lane DMA plan: 2560 pieces as five batched KiB per "virtual wave" v, pieces 2560..2583 as one more DMA of wave 0.
    //      FW_DMA_OLD_ONLY: the older wave of every SIMD (waves 0-3) issues the batches of virtual waves 2w and 2w+1 and the
    //      younger ones none: the younger wave is the critical path of every item, the older one waits at the barrier a
    //      quarter of the time. -----------------------------------------------------------------------------------------------
#ifndef FW_DMA_OLD_ONLY
#define FW_DMA_OLD_ONLY 0
#endif
    constexpr int NBB = FW_DMA_OLD_ONLY ? 2 : 1;                       // batches per issuing wave
    const bool issuer = FW_DMA_OLD_ONLY ? wave < NWAVES / 2 : true;    // wave-uniform
    auto vwave = [&](int bt) { return FW_DMA_OLD_ONLY ? 2 * wave + bt : wave; };
    auto piece_idx = [&](int bt, int i) { return i < ACT_ITERS ? (ACT_ITERS * vwave(bt) + i) * 64 + lane : ACT_ITERS * NWAVES * 64 + lane; };
    unsigned relb[NBB][ACT_ITERS];
    unsigned relb_x = 0;
    auto piece_off = [&](int idx) {
        const int row = idx / ROW_PIECES;
        const int rm = idx - row * ROW_PIECES;
        const int px = rm >> 2;
        const int s = (rm & 3) ^ halo_swz(px);
        return (unsigned)(((row * p.W + px) * p.in_cstride + s * 8) * 2);
    };
#pragma unroll
    for (int bt = 0; bt < NBB; ++bt)
#pragma unroll
        for (int i = 0; i < ACT_ITERS; ++i) relb[bt][i] = piece_off(piece_idx(bt, i)) + (unsigned)(4 - i) * 1024u;
    relb_x = piece_off(piece_idx(0, ACT_ITERS));
    const bool extra_lane = wave == 0 && lane < PS_EXTRA;
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
    const char* in = reinterpret_cast<const char*>(p.in);
    const char* wa_b = reinterpret_cast<const char*>(p.wpk_a);
    const char* wb_b = reinterpret_cast<const char*>(p.wpk_b);
    const unsigned lane16 = lane * 16;
    const long chunk_bytes = p.in_pstride * 2;

    // tile t: first conv_a row R0 and compute-region origin column ox (one pixel left of the first valid output)
    auto origin = [&](int t, int* R0, int* ox) {
        const int tx = t / tiles_y, ty = t - tx * tiles_y;
        *R0 = ty * PS_TH;
        *ox = tx * PS_TW - 1;
    };

    unsigned f_ok = 0;            // bit bt * ACT_ITERS + i: piece i of batch bt is inside the image; bit 2 * ACT_ITERS: the extra piece
    bool f_all = true;
    const char* f_src = nullptr;  // halo origin (image row R0 - 2, column ox - 1) of (tile f_t, chunk f_c)
    int f_t = t_begin, f_c = 0;
    auto plan_tile = [&]() {
        int R0, ox;
        origin(f_t, &R0, &ox);
        f_src = in + ((long)(R0 - 2) * p.W + (ox - 1)) * p.in_cstride * 2;
        f_ok = 0;
        auto inside = [&](int idx) {
            const int row = idx / ROW_PIECES;
            const int px = (idx - row * ROW_PIECES) >> 2;
            return idx < PS_PIECES && (unsigned)(R0 - 2 + row) < (unsigned)p.H && (unsigned)(ox - 1 + px) < (unsigned)p.W;
        };
#pragma unroll
        for (int bt = 0; bt < NBB; ++bt)
#pragma unroll
            for (int i = 0; i < ACT_ITERS; ++i)
                if (inside(piece_idx(bt, i))) f_ok |= 1u << (bt * ACT_ITERS + i);
        if (inside(piece_idx(0, ACT_ITERS))) f_ok |= 1u << (2 * ACT_ITERS);
        // the extra piece only counts on the lanes that issue it
        const unsigned need = ((1u << (NBB * ACT_ITERS)) - 1u) | (extra_lane ? 1u << (2 * ACT_ITERS) : 0u);
        f_all = __builtin_amdgcn_readfirstlane(__all((f_ok & need) == need)) != 0;
#ifdef FW_FORCE_SLOW_DMA  // timing experiment: every tile takes the per-piece border path
        f_all = false;
#endif
    };
    auto issue_act = [&](int stage) {
        if (f_c == 0) plan_tile();
        if (issuer) {
            const unsigned dst_x = lds_base + (unsigned)(stage * PS_REGION + ACT_ITERS * NWAVES * 64) * 16u;
#pragma unroll
            for (int bt = 0; bt < NBB; ++bt) {
                const unsigned dst = (unsigned)(stage * PS_REGION + ACT_ITERS * vwave(bt) * 64);
                if (f_all) {
                    glds16_batch_a(f_src, relb[bt], lds_base + (dst + 4 * 64) * 16u);
                } else {
#pragma unroll
                    for (int i = 0; i < ACT_ITERS; ++i)
                        glds16_v(((f_ok >> (bt * ACT_ITERS + i)) & 1u) ? f_src + (relb[bt][i] - (unsigned)(4 - i) * 1024u)
                                                                        : reinterpret_cast<const char*>(p.zeros),
                                 lds_base + (dst + i * 64) * 16u);
                }
            }
            if (extra_lane) {
                if (f_all)
                    glds16(f_src, relb_x, dst_x);
                else
                    glds16_v(((f_ok >> (2 * ACT_ITERS)) & 1u) ? f_src + relb_x : reinterpret_cast<const char*>(p.zeros), dst_x);
            }
        }
        if (++f_c == na) {
            f_c = 0;
            ++f_t;
        } else {
            f_src += chunk_bytes;
        }
    };
    auto issue_w = [&](int j, int ws) {
        if (!issuer) return;
#pragma unroll
        for (int bt = 0; bt < NBB; ++bt) {
            const int v = vwave(bt);
            const int half = v / (NWAVES / 2), k = v % (NWAVES / 2);
            if (half == 1 || j < na)
                issue_w_half((half ? wb_b : wa_b) + (size_t)j * (W_FRAGS * 1024),
                             lds_base + (unsigned)(SM::W_BASE + ws * SM::W_REGION + W_FRAGS * half * 64) * 16u, k, lane16);
        }
    };

    int rd_off[3][2];  // [dx][ph]: piece index of (halo row 2 * wave, px 16*ph + q + dx, slot sl)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int px = 16 * ph + q + dx;
            rd_off[dx][ph] = (RPW * wave) * ROW_PIECES + px * 4 + (sl ^ halo_swz(px));
        }
    auto widx = [](int tap, int w) { return w < 2 ? tap * 2 + w : W_FRAGS + tap * 2 + (w - 2); };

    f32x4 acc[RPW][NW][2];

    // ---- emit pieces (conv3x3_pair.hip has the lane map of the permlane swap) -----------------------------------------------
    const int ls = (sl & 1) ? 2 + (sl >> 1) : (sl >> 1);
    const unsigned st_off = (unsigned)((q * p.out_cstride + ls * 8) * 2);
    auto convert = [&](int w0, uint4 (&pk)[RPW][2]) {
#pragma unroll
        for (int row = 0; row < RPW; ++row)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const f32x4 va = lrelu4(w0 ? acc[row][2][ph] : acc[row][0][ph]);
                const f32x4 vb = lrelu4(w0 ? acc[row][3][ph] : acc[row][1][ph]);
                const uint2 pa = Op<T>::pack4(va[0], va[1], va[2], va[3]);
                const uint2 pb = Op<T>::pack4(vb[0], vb[1], vb[2], vb[3]);
                const u32x2 sx = __builtin_amdgcn_permlane16_swap(pa.x, pb.x, false, false);
                const u32x2 sy = __builtin_amdgcn_permlane16_swap(pa.y, pb.y, false, false);
                pk[row][ph] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
    };
    // the wave's two rows start at image row gy0; columns 1..30 of the region are valid outputs
    auto store_out = [&](const uint4 (&pk)[RPW][2], int gy0, int ox, T* plane, auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int row = 0; row < RPW; ++row) {
            const int gy = gy0 + row;
            if (!INTERIOR && !((unsigned)gy < (unsigned)p.H)) continue;  // wave-uniform
            char* rowbase = reinterpret_cast<char*>(plane) + ((long)gy * p.W + ox) * p.out_cstride * 2;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int cp = 16 * ph + q;
                if (cp >= 1 && cp <= PS_TW && (INTERIOR || (unsigned)(ox + cp) < (unsigned)p.W))
                    store16(rowbase + (long)(16 * ph) * p.out_cstride * 2 + st_off, pk[row][ph]);
            }
        }
    };
    // conv_a row cr of the tile -> x_a tile row cr + 2 (zero outside the image = conv_b's zero padding)
    auto write_xa = [&](const uint4 (&pk)[RPW][2], uint4* xa, int R0, int ox, auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int row = 0; row < RPW; ++row) {
            const int cr = RPW * wave + row;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int cp = 16 * ph + q, hp = cp + 1;
                uint4 v = pk[row][ph];
                if (!INTERIOR && !((unsigned)(R0 + cr) < (unsigned)p.H && (unsigned)(ox + cp) < (unsigned)p.W)) v = make_uint4(0, 0, 0, 0);
                xa[((cr + 2) * HALO_W + hp) * 4 + (ls ^ halo_swz(hp))] = v;
            }
        }
    };
    uint4* carry = lds + SM::CARRY;

    issue_w(0, 0);
    issue_act(0);

    int n = 0;   // item counter (weight stage = n & 1)
    int qd = 0;  // DMA'd chunks consumed
    for (int t = t_begin; t < t_hi; ++t) {
        const bool warm = t < t_lo;  // conv_a only, nothing stored
        int R0, ox;
        origin(t, &R0, &ox);
        const bool col_top = R0 == 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>((w < 2 ? p.bias_a : p.bias_b) + 16 * (w & 1) + 4 * sl);
#pragma unroll
            for (int row = 0; row < RPW; ++row)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) acc[row][w][ph] = bv;
        }
        FW_STAMP(4);  // tile set-up
        const int ipt = warm ? na : na + 1;
        // kind: 0 = shared chunk with both convs, 1 = shared chunk, conv_a only (warm-up), 2 = conv_b's x_a chunk
        auto run_item = [&](int j, auto kind_tag, bool wait_dma) {
            constexpr int KIND = decltype(kind_tag)::value;
            constexpr bool SHARED = KIND != 2;
            FW_STAMP(SHARED ? 1 : 2);  // the previous phase ends
            if (wait_dma) FW_WAIT_VMCNT(0);
            FW_STAMP(5);
            __syncthreads();
            FW_STAMP(0);
            const bool more = n + 1 < nitems;
            const int jn = (j + 1 == ipt) ? 0 : j + 1;
            const bool fetch = SHARED && more && (j + 1 < na || t + 1 < t_hi);
            const int fetch_stage = (qd + 1) & 1;
            auto dma_slot = [&](int d) {
                if (d == FW_DMA_SLOT_W) {
                    if (more) issue_w(jn, (n + 1) & 1);
                } else if (d == FW_DMA_SLOT_A) {
                    if (fetch) issue_act(fetch_stage);
                }
            };
            const uint4* a = lds + ((SHARED ? qd : qd - 1) & 1) * PS_REGION;
            const uint4* wl = lds + SM::W_BASE + (n & 1) * SM::W_REGION + lane;
            auto& slot_fn = dma_slot;
            if constexpr (KIND == 0)
                conv_item_lag<T, true>(acc, a, wl, rd_off, widx, slot_fn);
            else if constexpr (KIND == 1) {
                // warm-up: only the carry rows (conv_a rows 14, 15 = the last wave's) are wanted; the other waves just keep
                // the DMA stream going
                if (wave == NWAVES - 1) {
                    conv_item_lag<T, false>(acc, a, wl, rd_off, widx, slot_fn);
                } else {
                    slot_fn(FW_DMA_SLOT_W);
                    slot_fn(FW_DMA_SLOT_A);
                }
            }
            else
                conv_item<T, NW, 2>(acc, a, wl, rd_off, widx, slot_fn, [](const uint4 (&)[RPW][2]) {}, [](int) {});
            if (SHARED) ++qd;
            ++n;
        };
        bool wait_dma = t == t_begin;
#pragma clang loop unroll(disable)
        for (int j = 0; j < na; ++j) {
            if (warm)
                run_item(j, std::integral_constant<int, 1>{}, wait_dma);
            else
                run_item(j, std::integral_constant<int, 0>{}, wait_dma);
            wait_dma = true;
        }
        FW_STAMP(1);
        // conv_a done for this wave; the DMAs in flight are an item old: wait ahead of the stores (vmcnt counts stores too)
        FW_WAIT_VMCNT(0);
        FW_STAMP(5);
        // everything this tile reads, computes and stores inside the image?  (uniform)
        const bool interior = ox >= 0 && ox + TILE_W <= p.W && R0 >= 1 && R0 + TILE_H <= p.H;
        uint4 pk[RPW][2];
        if (!warm || wave == NWAVES - 1) convert(0, pk);
        if (!warm) {
            if (interior)
                store_out(pk, R0 + RPW * wave, ox, reinterpret_cast<T*>(p.out_a), std::true_type{});
            else
                store_out(pk, R0 + RPW * wave, ox, reinterpret_cast<T*>(p.out_a), std::false_type{});
        }
        FW_STAMP(6);      // convert + stores of x_a
        __syncthreads();  // every wave is done with the last chunk's stage: it becomes the x_a tile
        FW_STAMP(0);
        uint4* xa = lds + ((qd - 1) & 1) * PS_REGION;
        if (!warm || wave == NWAVES - 1) {
            if (interior)
                write_xa(pk, xa, R0, ox, std::true_type{});
            else
                write_xa(pk, xa, R0, ox, std::false_type{});
        }
        if (!warm) {
            // x_a rows R0 - 2, R0 - 1 -> tile rows 0, 1: from the tile above (carry), zero at the top of a column
            if (lane < PS_CARRY / NWAVES) {
                const int i = wave * (PS_CARRY / NWAVES) + lane;
                xa[i] = col_top ? make_uint4(0, 0, 0, 0) : carry[i];
            }
            FW_STAMP(3);  // x_a tile (own rows + carry rows) into LDS
            run_item(na, std::integral_constant<int, 2>{}, false);
            FW_STAMP(2);
        } else {
            __syncthreads();
            FW_STAMP(0);
        }
        // rows 16, 17 of the x_a tile (wave 7's own conv_a rows: its LDS writes are ordered before these reads) -> carry, after
        // the barrier that ends every wave's reading of the old carry
        if (wave == NWAVES - 1) {
#pragma unroll
            for (int k = 0; k < (PS_CARRY + 63) / 64; ++k) {
                const int i = k * 64 + lane;
                if (i < PS_CARRY) carry[i] = xa[16 * ROW_PIECES + i];
            }
        }
        FW_STAMP(3);  // carry rows
        if (!warm) {
            // conv_b done for this wave: rows R0 - 1 + (2w, 2w + 1).  In flight: the x_a stores (an item old), the next weights.
            FW_WAIT_VMCNT(0);
            FW_STAMP(5);
            convert(2, pk);
            if (interior)
                store_out(pk, R0 - 1 + RPW * wave, ox, reinterpret_cast<T*>(p.out_b), std::true_type{});
            else
                store_out(pk, R0 - 1 + RPW * wave, ox, reinterpret_cast<T*>(p.out_b), std::false_type{});
            FW_STAMP(6);  // convert + stores of x_b
        }
    }
    FW_STAMP_FLUSH(p.stamps);
}

// FW_PAIR_SLIDE=1 / 0 forces the sliding-window / the ring kernel; unset: the sliding window when a workgroup gets at least six
// tiles (its warm-up tile and the uneven last round weigh less and less: 0.9 % ahead at 720p, 3.5 % at 1080p; the ring kernel is
// 1.5 % ahead at 960x540, four tiles per workgroup).  Read per launch (a getenv is nothing next to a launch): tests flip it
// inside one process.
bool pair_slide_enabled(int H, int W, int num_cus) {
    if (const char* e = getenv("FW_PAIR_SLIDE")) return atoi(e) != 0;
    const long tiles = (long)((W + PS_TW - 1) / PS_TW) * ((H + PS_TH) / PS_TH);
    return tiles >= 6L * num_cus;
}

void launch_conv3x3_pair_slide(DType dt, const ConvPairParams& p, int num_cus, hipStream_t stream) {
    const int tiles = ((p.W + PS_TW - 1) / PS_TW) * ((p.H + PS_TH) / PS_TH);
    dim3 grid(tiles < num_cus ? tiles : num_cus), block(64 * NWAVES);
    if (dt == DT_BF16)
        hipLaunchKernelGGL((conv3x3_pair_slide_kernel<__bf16>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((conv3x3_pair_slide_kernel<_Float16>), grid, block, 0, stream, p);
    FW_HIP_CHECK(hipGetLastError());
}

}  // namespace fw
