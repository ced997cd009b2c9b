// The sliding-window conv pair (conv3x3_pair_slide.hip: x_a = lrelu(conv_a([x..])), x_b = lrelu(conv_b([x.., x_a])), reference
// src/framewright/processors/aesrgan_face.py:184-187) with ALL 32 columns of a tile valid.
//
// conv3x3_pair_slide.hip removed the vertical ring of the fused pair; the horizontal one stayed: both convs run on a 32-column
// region and conv_b is valid on 30 of them, because it needs x_a one column to either side (+6.7 % MACs on the four growth convs
// of a dense block, a tile pitch of 30 pixels = 1920-byte output rows that straddle 128-byte lines).  Here a tile is 16 x 32
// valid pixels of BOTH outputs:
//
//   conv_b columns  [ox, ox + 31]            needs x_a columns [ox - 1, ox + 32]
//   conv_a columns  [ox - 1, ox + 32] (34):  [ox, ox + 31] by the row-shaped MFMAs of the window kernel (pixel = lane & 15 along a
//                   half row), the two extra columns ox - 1 and ox + 32 by COLUMN-shaped MFMAs: lane & 15 = one of the tile's 16
//                   rows, the same A (weight) fragment the wave already holds for that tap - 9 more MFMAs per shared chunk for each
//                   of four waves (one per SIMD: wave e < 4 owns column e & 1, output-channel tile e >> 1), +3.1 % instead of +6.7 %
//   input halo      columns [ox - 2, ox + 33] (36), rows [R0 - 2, R0 + 16] (19)
//
// LDS image of a 32-channel chunk: [19 rows][36 px][4 slots] + ONE pad piece per row (145 pieces): the column-shaped B fragments
// read 16 rows of one pixel with a stride of 145 pieces = 1 bank position (mod 16), so they are spread over the banks like the
// row-shaped ones (a stride of 144 would put all 16 rows on one).  The x_a tile uses the same layout (x_a column c at px c - ox + 2),
// so one set of fragment offsets serves both.  160 KiB of LDS holds two such stages + two weight stages only without the carry
// buffer: the two x_a rows a tile hands to the next one stay in REGISTERS (37 pieces per wave, one uint4 per lane).
// Per-pixel accumulation order is that of the ring and window kernels (chunks in order, taps dx-major), so frames are bit-identical.
#include <cstdlib>
#include <type_traits>
#include "fw_internal.h"
#include "conv_common.h"

#ifndef FW_DMA_SLOT_W
#define FW_DMA_SLOT_W 2
#define FW_DMA_SLOT_A 6
#endif

namespace fw {

constexpr int P32_TH = TILE_H;                   // 16 rows per step
constexpr int P32_TW = TILE_W;                   // 32 valid pixels per tile row
constexpr int P32_ROWS = TILE_H + 3;             // 19 halo rows
constexpr int P32_COLS = TILE_W + 4;             // 36 halo pixels
constexpr int P32_ROWP = P32_COLS * 4 + 1;       // 145 pieces per halo row (one pad piece)
constexpr int P32_PIECES = P32_ROWS * P32_ROWP;  // 2755
constexpr int P32_REGION = P32_PIECES;           // pieces per activation stage
constexpr int P32_SKIP = P32_PIECES - 43 * 64;    // 3: the stage's first three pieces (row 0, px 0: read by nobody) are never loaded,
                                                  // so that the other 2752 are exactly 43 wave-instructions: waves 0-3 and 7 issue
                                                  // five of them as one batch, waves 4-6 six (one M0 write per wave either way)
constexpr int P32_CARRY = 2 * P32_ROWP;          // two x_a rows = 290 pieces
constexpr int P32_CARRY_PW = (P32_CARRY + NWAVES - 1) / NWAVES;   // 37 per wave

struct Slide32Smem {
    static constexpr int W_REGION = 2 * W_FRAGS * 64;
    static constexpr int W_BASE = 2 * P32_REGION;
    static constexpr int TOTAL = W_BASE + 2 * W_REGION;   // 10118 pieces = 161888 bytes
};
static_assert(NWAVES == 8 && RPW == 2, "written for 8 waves of 2 rows");
static_assert(P32_SKIP >= 0 && P32_SKIP <= 4 && 5 * 5 + 3 * 6 == 43 && P32_CARRY_PW <= 64, "DMA plan");
static_assert(Slide32Smem::TOTAL * 16 <= 160 * 1024, "LDS");

// acc += A * B when `on` (wave-uniform) is non-zero, with the jump inside the statement: no control flow that hipcc can see.  The
// operands come from LDS reads (the compiler waits for them ahead of the statement) and the accumulator is next touched tiles later,
// so none of the MFMA's software wait states is at stake.
template <typename T>
__device__ __forceinline__ void mfma16_if(f32x4& acc, const uint4& a4, const uint4& b4, int on) {
    const nt_u32x4 a = {a4.x, a4.y, a4.z, a4.w}, b = {b4.x, b4.y, b4.z, b4.w};   // (a HIP uint4 is a struct: no register constraint takes it)
    on = __builtin_amdgcn_readfirstlane(on);   // provably scalar, even where hipcc has parked the flag in a vector register
    if constexpr (std::is_same<T, __bf16>::value)
        asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 .Lfw_skip_%=\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n.Lfw_skip_%=:"
                     : "+v"(acc) : "v"(a), "v"(b), "s"(on) : "scc");
    else
        asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 .Lfw_skip_%=\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n.Lfw_skip_%=:"
                     : "+v"(acc) : "v"(a), "v"(b), "s"(on) : "scc");
}

// One shared-chunk item with conv_b one row behind conv_a (conv3x3_pair_slide.hip: conv_item_lag), plus the extra conv_a column of
// a wave that owns one (xon): its B fragment for tap (dy, dx) is piece xcol + dy * P32_ROWP + xdx[dx] of the stage (16 rows of one
// halo pixel), its A fragment the one the row-shaped MFMAs of output-channel tile xct use at that tap.
// Only waves 0-3 own an extra column, and the item must not branch on that in a way hipcc sees: a wave-uniform branch per step made
// its waitcnt pass drain the LDS queue at every join (the item lost its read-ahead: +3.4 ms per frame), two whole variants of the
// item under one branch spilled 44-126 registers.  So the two fragment reads are predicated per lane (xlane: true in every lane of
// waves 0-3 - to the compiler a divergent condition, i.e. an EXEC mask around one ds_read, no branch) and the MFMA sits in an asm
// statement that jumps over it when xon == 0 (mfma16_if).  wlx = wl + 64 * xct: the extra column's own A fragment (tile xct of
// conv_a) is read a step ahead like the others, at a compile-time offset.  XCODE = false compiles all of it out (warm-up items).
template <typename T, bool WITH_B, bool XCODE, typename WIdx, typename Slot>
__device__ __forceinline__ void conv_item_lag32(f32x4 (&acc)[RPW][4][2], f32x4& accx, const uint4* a, const uint4* wl, const uint4* wlx,
                                                const int (&rd_off)[3][2], const int (&xoff)[3], bool xlane, int xon, WIdx widx, Slot dma_slot) {
    constexpr int NWU = WITH_B ? 4 : 2;
    constexpr int NK = 9 * NWU;
    constexpr int RING = 3;
    constexpr int ROWP = P32_ROWP;
    uint4 xr[5][2];
    uint4 wf[RING];
    uint4 xc = make_uint4(0, 0, 0, 0);   // the extra column's B fragment of the current tap: re-read once its MFMA has issued
    uint4 wx = make_uint4(0, 0, 0, 0);   // ... and its A fragment
    auto tile_of = [](int w) { return WITH_B ? ((w + 2) & 3) : w; };
    auto load_w = [&](int k) {
        const int t = k / NWU, w = k - t * NWU;
        const int dx = t / 3, dy = t - 3 * dx;
        wf[k % RING] = wl[widx(dy * 3 + dx, tile_of(w)) * 64];
    };
    auto load_row = [&](int h, int dx) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) xr[h][ph] = a[h * ROWP + rd_off[dx][ph]];
    };
#pragma unroll
    for (int h = 0; h < 5; ++h) load_row(h, 0);
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) load_w(k);
    if constexpr (XCODE) { if (xlane) xc = a[xoff[0]]; }   // tap (dy 0, dx 0)
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
            if constexpr (XCODE) {
                if (w == 0) { if (xlane) wx = wlx[widx(dy * 3 + dx, 0) * 64]; }
                if (w == NWU - 1) mfma16_if<T>(accx, wx, xc, xon);
            }
            FW_SB();
#pragma unroll
            for (int d = 0; d < 4 / NWU; ++d) dma_slot((t * NWU + w) * (4 / NWU) + d);
            FW_SB();
            if (dx < 2 && dy == 2 && w == (WITH_B ? 1 : 0)) load_row(2, dx + 1);
        }
        if (t + 1 < 9) {
            const int dxn = (t + 1) / 3, dyn = (t + 1) - 3 * dxn;
            if constexpr (XCODE) { if (xlane) xc = a[xoff[dxn] + dyn * ROWP]; }
        }
    }
}

// A shared chunk of a warm-up tile for a wave that owns an extra column and no rows: its nine column-shaped MFMAs only.
template <typename T, typename WIdx>
__device__ __forceinline__ void conv_item_xcol(f32x4& accx, const uint4* a, const uint4* wlx, const int (&xoff)[3], WIdx widx) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dx = t / 3, dy = t - 3 * dx;
        const uint4 wx = wlx[widx(dy * 3 + dx, 0) * 64];
        const uint4 xc = a[xoff[dx] + dy * P32_ROWP];
        accx = Op<T>::mfma16(wx, xc, accx);
    }
}

template <typename T>
__global__ __launch_bounds__(64 * NWAVES, WAVES_PER_SIMD) void conv3x3_pair_slide32_kernel(const ConvPairParams p) {
    using SM = Slide32Smem;
    __shared__ __attribute__((aligned(16))) uint4 lds[SM::TOTAL];
    constexpr int NW = 4;
    constexpr int ROWP = P32_ROWP;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int q = lane & 15;
    const int sl = lane >> 4;

    const int NB = gridDim.x;
    const int xcd = blockIdx.x & 7;
    const int qn = NB >> 3, rn = NB & 7;
    const int lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int tiles_x = (p.W + P32_TW - 1) / P32_TW;
    const int tiles_y = (p.H + P32_TH) / P32_TH;  // the last conv_b row H - 1 = 16 ty + 14 at the latest
    const int ntiles = tiles_x * tiles_y;         // column-major: t = tx * tiles_y + ty
    const int t_lo = (int)((long)lb * ntiles / NB);
    const int t_hi = (int)((long)(lb + 1) * ntiles / NB);
    if (t_lo >= t_hi) return;
    const int needs_warm = (t_lo % tiles_y) != 0;  // the run starts below the top of a column: conv_a of the tile above first
    const int t_begin = t_lo - needs_warm;
    const int na = p.na;
    const int nitems = (t_hi - t_lo) * (na + 1) + needs_warm * na;

    // ---- per-lane DMA plan: pieces P32_SKIP .. 2754 as 43 KiB: wave w issues KiB [kib0, kib0 + nb) as ONE batch (nb = 6 for waves 4-6,
    //      else 5).  A row's pad piece (rm == 144) repeats the piece before it: never read. --------------------------------------
    const bool six = wave >= 4 && wave < 7;                                   // wave-uniform
    const int kib0 = 5 * wave + (wave > 4 ? (wave < 7 ? wave - 4 : 3) : 0);   // 0, 5, 10, 15, 20, 26, 32, 38
    constexpr int NBMAX = ACT_ITERS + 1;
    auto piece_idx = [&](int i) { return P32_SKIP + (kib0 + i) * 64 + lane; };
    auto piece_off = [&](int idx) {
        const int row = idx / ROWP;
        int rm = idx - row * ROWP;
        rm = rm < ROWP - 1 ? rm : ROWP - 2;
        const int px = rm >> 2;
        const int s = (rm & 3) ^ halo_swz(px);
        return (unsigned)(((row * p.W + px) * p.in_cstride + s * 8) * 2);
    };
    unsigned relb[NBMAX];
#pragma unroll
    for (int i = 0; i < NBMAX; ++i) relb[i] = (i < ACT_ITERS || six) ? piece_off(piece_idx(i)) + (unsigned)((4 - i) * 1024) : 0u;
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
    const char* in = reinterpret_cast<const char*>(p.in);
    const char* wa_b = reinterpret_cast<const char*>(p.wpk_a);
    const char* wb_b = reinterpret_cast<const char*>(p.wpk_b);
    const unsigned lane16 = lane * 16;
    const long chunk_bytes = p.in_pstride * 2;

    // tile t: first conv_a row R0 and first valid column ox
    auto origin = [&](int t, int* R0, int* ox) {
        const int tx = t / tiles_y, ty = t - tx * tiles_y;
        *R0 = ty * P32_TH;
        *ox = tx * P32_TW;
    };

    unsigned f_ok = 0;            // bit i: piece i of the wave's batch is inside the image
    bool f_all = true;
    const char* f_src = nullptr;  // halo origin (image row R0 - 2, column ox - 2) of (tile f_t, chunk f_c)
    int f_t = t_begin, f_c = 0;
    auto plan_tile = [&]() {
        int R0, ox;
        origin(f_t, &R0, &ox);
        f_src = in + ((long)(R0 - 2) * p.W + (ox - 2)) * p.in_cstride * 2;
        f_ok = 0;
        auto inside = [&](int idx) {
            const int row = idx / ROWP;
            int rm = idx - row * ROWP;
            rm = rm < ROWP - 1 ? rm : ROWP - 2;
            const int px = rm >> 2;
            return (unsigned)(R0 - 2 + row) < (unsigned)p.H && (unsigned)(ox - 2 + px) < (unsigned)p.W;
        };
#pragma unroll
        for (int i = 0; i < NBMAX; ++i)
            if ((i < ACT_ITERS || six) && inside(piece_idx(i))) f_ok |= 1u << i;
        const unsigned need = six ? (1u << NBMAX) - 1u : (1u << ACT_ITERS) - 1u;
        f_all = __builtin_amdgcn_readfirstlane(__all((f_ok & need) == need)) != 0;
    };
    auto issue_act = [&](int stage) {
        if (f_c == 0) plan_tile();
        const unsigned dst = (unsigned)(stage * P32_REGION + P32_SKIP + kib0 * 64);   // the wave's first piece
        if (f_all) {
            if (six)
                glds16_batch_a6(f_src, relb, lds_base + (dst + 4 * 64) * 16u);
            else
                glds16_batch_a(f_src, relb, lds_base + (dst + 4 * 64) * 16u);
        } else {
#pragma unroll
            for (int i = 0; i < NBMAX; ++i)
                if (i < ACT_ITERS || six)
                    glds16_v(((f_ok >> i) & 1u) ? f_src + (relb[i] - (unsigned)((4 - i) * 1024)) : reinterpret_cast<const char*>(p.zeros),
                             lds_base + (dst + i * 64) * 16u);
        }
        if (++f_c == na) {
            f_c = 0;
            ++f_t;
        } else {
            f_src += chunk_bytes;
        }
    };
    auto issue_w = [&](int j, int ws) {
        const int half = wave / (NWAVES / 2), k = wave % (NWAVES / 2);
        if (half == 1 || j < na)
            issue_w_half((half ? wb_b : wa_b) + (size_t)j * (W_FRAGS * 1024),
                         lds_base + (unsigned)(SM::W_BASE + ws * SM::W_REGION + W_FRAGS * half * 64) * 16u, k, lane16);
    };

    int rd_off[3][2];  // [dx][ph]: piece index of (halo row 2 * wave, column ox + 16 ph + q + dx - 1 = px 16 ph + q + dx + 1, slot sl)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int px = 16 * ph + q + dx + 1;
            rd_off[dx][ph] = (RPW * wave) * ROWP + px * 4 + (sl ^ halo_swz(px));
        }
    auto widx = [](int tap, int w) { return w < 2 ? tap * 2 + w : W_FRAGS + tap * 2 + (w - 2); };

    // ---- the extra conv_a column of waves 0-3: column ox - 1 (xcs 0) or ox + 32 (xcs 1), output-channel tile xct; lane = (row q of
    //      the tile, slot sl).  Tap (dy, dx) reads halo row q + dy + 1, px (xcs ? 33 : 0) + dx. ------------------------------------
#ifdef FW_P32_NOX   // timing only: no column-shaped MFMAs (wrong pixels in a tile's first and last column of x_b)
    const bool xon = false;
#else
    const bool xon = wave < 4;
#endif
    const int xcs = wave & 1, xct = (wave >> 1) & 1;
    const bool xlane = xon && tid < 64 * NWAVES;   // == xon in every lane; kept per-lane so that hipcc predicates instead of branching
    int xoff[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int px = (xcs ? 33 : 0) + dx;
        xoff[dx] = (q + 1) * ROWP + px * 4 + (sl ^ halo_swz(px));
    }

    f32x4 acc[RPW][NW][2];
    f32x4 accx;

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
    // the wave's two rows start at image row gy0; all 32 columns of the tile are valid outputs: whole 2-KiB rows
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
                if (INTERIOR || ox + cp < p.W) store16(rowbase + (long)(16 * ph) * p.out_cstride * 2 + st_off, pk[row][ph]);
            }
        }
    };
    // conv_a row cr, column ox + cp of the tile -> x_a tile row cr + 2, px cp + 2 (zero outside the image = conv_b's zero padding)
    auto write_xa = [&](const uint4 (&pk)[RPW][2], uint4* xa, int R0, int ox, auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int row = 0; row < RPW; ++row) {
            const int cr = RPW * wave + row;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int cp = 16 * ph + q, hp = cp + 2;
                uint4 v = pk[row][ph];
                if (!INTERIOR && !((unsigned)(R0 + cr) < (unsigned)p.H && ox + cp < p.W)) v = make_uint4(0, 0, 0, 0);
                xa[(cr + 2) * ROWP + hp * 4 + (ls ^ halo_swz(hp))] = v;
            }
        }
    };
    // the extra column: lane (row q, channels 16 xct + 4 sl .. + 3) -> 8 bytes of x_a tile row q + 2, px 1 or 34
    auto write_xa_col = [&](uint4* xa, int R0, int ox) {
        const f32x4 v = lrelu4(accx);
        uint2 h = Op<T>::pack4(v[0], v[1], v[2], v[3]);
        const int gx = xcs ? ox + P32_TW : ox - 1;
        if (!((unsigned)(R0 + q) < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)) h = make_uint2(0u, 0u);
        const int hp = xcs ? P32_TW + 2 : 1;
        char* dst = reinterpret_cast<char*>(xa + (q + 2) * ROWP + hp * 4 + ((2 * xct + (sl >> 1)) ^ halo_swz(hp))) + 8 * (sl & 1);
        *reinterpret_cast<uint2*>(dst) = h;
    };
    // the two x_a rows this tile hands to the next one: piece wave * 37 + lane of tile rows 16, 17
    uint4 carry = make_uint4(0, 0, 0, 0);
    const int ci = wave * P32_CARRY_PW + lane;
    const bool carry_lane = lane < P32_CARRY_PW && ci < P32_CARRY;

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
        accx = *reinterpret_cast<const f32x4*>(p.bias_a + 16 * xct + 4 * sl);
        const int ipt = warm ? na : na + 1;
        // kind: 0 = shared chunk with both convs, 1 = shared chunk, conv_a only (warm-up), 2 = conv_b's x_a chunk
        auto run_item = [&](int j, auto kind_tag, bool wait_dma) {
            constexpr int KIND = decltype(kind_tag)::value;
            constexpr bool SHARED = KIND != 2;
            if (wait_dma) FW_WAIT_VMCNT(0);
            __syncthreads();
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
            const uint4* a = lds + ((SHARED ? qd : qd - 1) & 1) * P32_REGION;
            const uint4* wl = lds + SM::W_BASE + (n & 1) * SM::W_REGION + lane;
            auto& slot_fn = dma_slot;
            const uint4* wlx = wl + 64 * xct;
            if constexpr (KIND == 0) {
                conv_item_lag32<T, true, true>(acc, accx, a, wl, wlx, rd_off, xoff, xlane, xon ? 1 : 0, widx, slot_fn);
            } else if constexpr (KIND == 1) {
                // warm-up: only the carry rows (conv_a rows 14, 15 = the last wave's, and the same rows of the two extra columns) are
                // wanted; the other waves just keep the DMA stream going
                if (wave == NWAVES - 1) {
                    conv_item_lag32<T, false, false>(acc, accx, a, wl, wlx, rd_off, xoff, false, 0, widx, slot_fn);
                } else {
                    slot_fn(FW_DMA_SLOT_W);
                    slot_fn(FW_DMA_SLOT_A);
                    if (xon) conv_item_xcol<T>(accx, a, wlx, xoff, widx);
                }
            }
            else
                conv_item_rp<P32_ROWP, T, NW, 2>(acc, a, wl, rd_off, widx, slot_fn, [](const uint4 (&)[RPW][2]) {}, [](int) {});
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
        // conv_a done for this wave; the DMAs in flight are an item old: wait ahead of the stores (vmcnt counts stores too)
        FW_WAIT_VMCNT(0);
        // everything this tile computes and stores inside the image?  (uniform; the extra columns check for themselves)
        const bool interior = ox + P32_TW <= p.W && R0 >= 1 && R0 + P32_TH <= p.H;
        uint4 pk[RPW][2];
        if (!warm || wave == NWAVES - 1) convert(0, pk);
        if (!warm) {
            if (interior)
                store_out(pk, R0 + RPW * wave, ox, reinterpret_cast<T*>(p.out_a), std::true_type{});
            else
                store_out(pk, R0 + RPW * wave, ox, reinterpret_cast<T*>(p.out_a), std::false_type{});
        }
        __syncthreads();  // every wave is done with the last chunk's stage: it becomes the x_a tile
        uint4* xa = lds + ((qd - 1) & 1) * P32_REGION;
        if (!warm || wave == NWAVES - 1) {
            if (interior)
                write_xa(pk, xa, R0, ox, std::true_type{});
            else
                write_xa(pk, xa, R0, ox, std::false_type{});
        }
        if (xon) write_xa_col(xa, R0, ox);
        if (!warm) {
            // x_a rows R0 - 2, R0 - 1 -> tile rows 0, 1: from the tile above (carry registers), zero at the top of a column
            if (carry_lane) xa[ci] = col_top ? make_uint4(0, 0, 0, 0) : carry;
            run_item(na, std::integral_constant<int, 2>{}, false);
        } else {
            __syncthreads();
        }
        // rows 16, 17 of the x_a tile -> carry registers: behind the barrier that made every wave's x_a writes visible, ahead of the
        // next item's barrier, behind which the DMAs refill this stage
        if (carry_lane) carry = xa[16 * ROWP + ci];
        if (!warm) {
            // conv_b done for this wave: rows R0 - 1 + (2w, 2w + 1).  In flight: the x_a stores (an item old), the next weights.
            FW_WAIT_VMCNT(0);
            convert(2, pk);
            if (interior)
                store_out(pk, R0 - 1 + RPW * wave, ox, reinterpret_cast<T*>(p.out_b), std::true_type{});
            else
                store_out(pk, R0 - 1 + RPW * wave, ox, reinterpret_cast<T*>(p.out_b), std::false_type{});
        }
    }
}

// Measured (profiles/r03_ab/pair_slide32.txt, one box per table, f16 1080p frame): with the extra columns switched off (wrong pixels,
// -DFW_P32_NOX) the 32-column geometry is worth 0.8 - 1.5 ms of a 68.6 - 69.5 ms frame; the 36 column-shaped MFMAs per chunk cost 1.7 ms
// (3.4 ms before they were made invisible to hipcc's control flow): the frame is 0.9 ms SLOWER than with the 30-column kernel.  The
// chip runs these kernels at its power cap, so an executed MFMA costs time wherever it is placed - the slack of the waves that own
// a column does not hide it - and the ring is only 3.6 % of the pair's MACs cheaper to begin with.  The kernel is kept as an
// independent third implementation of the pair for the bit-identity tests and is OFF by default: FW_PAIR_SLIDE=2 (or
// FW_PAIR_SLIDE32=1) selects it.
bool pair_slide32_enabled() {
    if (const char* e = getenv("FW_PAIR_SLIDE")) return atoi(e) == 2;
    if (const char* e = getenv("FW_PAIR_SLIDE32")) return atoi(e) != 0;
    return false;
}

void launch_conv3x3_pair_slide32(DType dt, const ConvPairParams& p, int num_cus, hipStream_t stream) {
    const int tiles = ((p.W + P32_TW - 1) / P32_TW) * ((p.H + P32_TH) / P32_TH);
    dim3 grid(tiles < num_cus ? tiles : num_cus), block(64 * NWAVES);
    if (dt == DT_BF16)
        hipLaunchKernelGGL((conv3x3_pair_slide32_kernel<__bf16>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((conv3x3_pair_slide32_kernel<_Float16>), grid, block, 0, stream, p);
    FW_HIP_CHECK(hipGetLastError());
}

}  // namespace fw
