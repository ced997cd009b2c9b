// conv5 of a residual dense block (192 -> 64 channels + the split-trunk residuals, EPI_RESIDUAL_SPLIT of conv3x3_mfma.hip; reference
// src/framewright/processors/aesrgan_face.py:188-189, 204) as a ROW-WISE WINOGRAD F(2, 3) on the matrix cores: two thirds of the MFMAs of
// the direct form.  f16 operands only.
//
// Per output row y and column pair (2j, 2j + 1), 32 input channels at a time:
//     m_f[cout] += sum_dy U[f][dy][cout][cin] * V_f[cin](input row y + dy - 1),   f = 0 .. 3
//     V = B^T d:  (d0 - d2, d1 + d2, d2 - d1, d1 - d3),  d_k = input column 2j - 1 + k      (v_pk_add_f16 on the raw fragments)
//     U = G g:    (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2) of tap row dy, summed in fp32 on the host and rounded to f16 once
//     y(2j) = m0 + m1 + m2,   y(2j + 1) = m1 - m2 - m3
// tools/winograd1d_kloop.py measured the K loop on static data: 3.9k against the direct item's 5.0k cycles (profiles/r03_winograd1d_kloop.json).
//
// Mapping: the direct kernels' - persistent 512-thread workgroups, one 16 x 32 tile at a time, wave w = output rows 2w, 2w + 1 and all four
// 16-channel output tiles; lane = (column pair j = lane & 15, 8-channel slot sl = lane >> 4).  B fragment of frequency f = V_f of the lane's
// pair, A fragment = U[f][dy][tile] (48 per chunk, each feeding the wave's two rows), D = 4 output channels of the frequency-f accumulator:
// 32 accumulator tiles (128 registers) + the 16 transformed fragments of the wave's four halo rows (64 registers).
// LDS: ONE activation stage (the raw 18 x 34 halo tile is only read at the top of an item, into the V fragments: the next item's DMA goes
// out behind a second barrier right after) + two weight stages of 48 KiB = 134 KiB.  The raw tile is stored de-interleaved -
// [row][column parity][17 columns][4 slots] - so that the lanes of a read (columns 2j + k) are 64 bytes apart like the direct kernel's.
// Residuals: the conv's own input chunks 0, 1 (x hi) come from the raw centre fragments while they are in registers, residual planes from
// HBM as in the direct kernel; a pixel of the pair's first / second column enters frequency 0 with + s I / frequency 3 with - s I
// (y(2j) takes + m0, y(2j + 1) takes - m3).  The bias sits in m1, which both outputs take with + 1.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "fw_internal.h"
#include "conv_common.h"

namespace fw {

constexpr int WN_ROWP = 2 * 17 * 4;                 // pieces per halo row of the de-interleaved image (136)
constexpr int WN_ACT = HALO_H * WN_ROWP;            // 2448 pieces
constexpr int WN_ACT_REGION = ACT_INSTR * 64;       // 2560: the batched DMAs write whole KiB
constexpr int WN_WFR = 3 * 4 * 4;                   // weight fragments per chunk: [dy][f][tile]
constexpr int WN_W_REGION = WN_WFR * 64;            // 3072 pieces = 48 KiB
constexpr int WN_TOTAL = WN_ACT_REGION + 2 * WN_W_REGION;
static_assert(NWAVES == 8 && RPW == 2 && WN_TOTAL * 16 <= 160 * 1024, "written for 8 waves of 2 rows");

typedef _Float16 wn_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 wn_sub(uint4 a, uint4 b) {
    uint4 r;
    r.x = __builtin_bit_cast(unsigned, __builtin_bit_cast(wn_h2, a.x) - __builtin_bit_cast(wn_h2, b.x));
    r.y = __builtin_bit_cast(unsigned, __builtin_bit_cast(wn_h2, a.y) - __builtin_bit_cast(wn_h2, b.y));
    r.z = __builtin_bit_cast(unsigned, __builtin_bit_cast(wn_h2, a.z) - __builtin_bit_cast(wn_h2, b.z));
    r.w = __builtin_bit_cast(unsigned, __builtin_bit_cast(wn_h2, a.w) - __builtin_bit_cast(wn_h2, b.w));
    return r;
}
__device__ __forceinline__ uint4 wn_add(uint4 a, uint4 b) {
    uint4 r;
    r.x = __builtin_bit_cast(unsigned, __builtin_bit_cast(wn_h2, a.x) + __builtin_bit_cast(wn_h2, b.x));
    r.y = __builtin_bit_cast(unsigned, __builtin_bit_cast(wn_h2, a.y) + __builtin_bit_cast(wn_h2, b.y));
    r.z = __builtin_bit_cast(unsigned, __builtin_bit_cast(wn_h2, a.z) + __builtin_bit_cast(wn_h2, b.z));
    r.w = __builtin_bit_cast(unsigned, __builtin_bit_cast(wn_h2, a.w) + __builtin_bit_cast(wn_h2, b.w));
    return r;
}

// Six weight KiB of one wave as one batch (one M0 write): the fifth piece is M0's, the sixth one KiB above it.
__device__ __forceinline__ void wn_glds_w6(const void* sbase, unsigned voff, unsigned lds_piece4) {
    const unsigned long long b = (unsigned long long)sbase;
    const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b);
    lds_piece4 = __builtin_amdgcn_readfirstlane(lds_piece4);
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:-4096\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:-3072\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:-2048\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:-1024\n\t"
        "global_load_lds_dwordx4 %0, %1\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:1024"
        :
        : "v"(voff), "s"(sb), "s"(lds_piece4)
        : "memory");
}

// PLANES: with residual planes from HBM (rdb3's conv5); without, their 16 fragment registers do not exist (rdb1 / rdb2: two thirds of the launches)
// STORE: the plain epilogue instead of the split trunk's - typed output of act(conv + bias) (conv_hr of the RRDBNet tail at 8K)
template <bool PLANES, bool STORE>
__global__ __launch_bounds__(64 * NWAVES, WAVES_PER_SIMD) void conv3x3_wino_split_kernel(const ConvParams p) {
    using T = _Float16;
    __shared__ __attribute__((aligned(16))) uint4 lds[WN_TOTAL];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 15;    // column pair of the tile (columns 2j, 2j + 1); output channel within a tile for A fragments
    const int sl = lane >> 4;   // 8-channel slot of the chunk; 4-channel group of a D tile

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
    const int nch = p.cin_chunks;
    const int nitems = (t_hi - t_lo) * nch;

    // ---- activation DMA plan: piece idx of the de-interleaved image -> (halo row, halo column p = 2 i + parity, slot s); pieces behind
    //      the image (the batches write whole KiB) repeat the image's first ones ------------------------------------------------------------
    unsigned relb[ACT_ITERS];
    auto piece_pos = [&](int i, int* row, int* px, int* s) {
        int idx = (ACT_ITERS * wave + i) * 64 + lane;
        if (idx >= WN_ACT) idx -= WN_ACT;
        const int rw = idx / WN_ROWP;
        const int rm = idx - rw * WN_ROWP;
        const int par = rm >= 68 ? 1 : 0;
        const int q = rm - 68 * par;
        const int ci = q >> 2;
        *row = rw;
        *px = 2 * ci + par;
        *s = (q & 3) ^ halo_swz(ci);
    };
#pragma unroll
    for (int i = 0; i < ACT_ITERS; ++i) {
        int row, px, s;
        piece_pos(i, &row, &px, &s);
        relb[i] = (unsigned)(((row * p.W + px) * p.in_cstride + s * 8) * 2) + (unsigned)(4 - i) * 1024u;
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
    const char* in = reinterpret_cast<const char*>(p.in);
    const char* w_b = reinterpret_cast<const char*>(p.wpk);
    const unsigned lane16 = lane * 16;
    const long chunk_bytes = p.in_pstride * 2;

    unsigned a_ok = 0;
    bool a_all = true;
    const char* a_src = nullptr;
    int a_t = t_lo, a_c = 0;
    auto plan_tile = [&]() {
        int aty, atx;
        tile_pos(a_t, tiles_x, tiles_y, &aty, &atx);
        const int ty0 = aty * TILE_H, tx0 = atx * TILE_W;
        a_src = in + ((long)(ty0 - 1) * p.W + (tx0 - 1)) * p.in_cstride * 2;
        a_ok = 0;
#pragma unroll
        for (int i = 0; i < ACT_ITERS; ++i) {
            int row, px, s;
            piece_pos(i, &row, &px, &s);
            const int gy = ty0 - 1 + row, gx = tx0 - 1 + px;
            if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) a_ok |= 1u << i;
        }
        a_all = __builtin_amdgcn_readfirstlane(__all(a_ok == (1u << ACT_ITERS) - 1u)) != 0;
    };
    auto issue_act = [&]() {
        if (a_c == 0) plan_tile();
        const unsigned dst = (unsigned)(ACT_ITERS * wave * 64);
        if (a_all) {
            glds16_batch_a(a_src, relb, lds_base + (dst + 4 * 64) * 16u);
        } else {
#pragma unroll
            for (int i = 0; i < ACT_ITERS; ++i)
                glds16_v(((a_ok >> i) & 1u) ? a_src + (relb[i] - (unsigned)(4 - i) * 1024u) : reinterpret_cast<const char*>(p.zeros),
                         lds_base + (dst + i * 64) * 16u);
        }
        if (++a_c == nch) {
            a_c = 0;
            ++a_t;
        } else {
            a_src += chunk_bytes;
        }
    };
    // weights of chunk c into stage ws: wave w the KiB 6w .. 6w + 5
    auto issue_w = [&](int c, int ws) {
        wn_glds_w6(w_b + (size_t)c * (WN_WFR * 1024) + (6 * wave + 4) * 1024, lane16,
                   lds_base + (unsigned)(WN_ACT_REGION + ws * WN_W_REGION + (6 * wave + 4) * 64) * 16u);
    };

    int rd[4];   // piece offset (within a halo row) of column 2j + k: halo column index = 2j + k (the halo starts one column left of the tile)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ci = j + (k >> 1);
        rd[k] = (k & 1) * 68 + ci * 4 + (sl ^ halo_swz(ci));
    }
    // identity A fragments: tile ctl of a 32-channel plane, row = cout 16 ctl + j, k = 8 sl + e -> 1 where cout == k
    unsigned id_mask = 0;
#pragma unroll
    for (int ctl = 0; ctl < 2; ++ctl)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (16 * ctl + j == 8 * sl + e) id_mask |= 1u << (8 * ctl + e);

    f32x4 acc[RPW][4][4];   // [row][frequency][16-channel tile]

    // acc[row][f][2 c2 + ctl] += sc * xf[row] (32 channels of the wave's two rows, one column of every pair): identity A fragments
    auto add_identity = [&](const uint4 (&xf)[RPW], float sc, int c2, auto freq_tag) {
        constexpr int F = decltype(freq_tag)::value;
#pragma unroll
        for (int ctl = 0; ctl < 2; ++ctl) {
            // (the empty asm keeps hipcc from building every identity fragment of the kernel once and holding them all - they are loop
            // invariant: 16 to 64 registers at the pressure peak)
            asm volatile("" : "+v"(sc));
            float e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = ((id_mask >> (8 * ctl + k)) & 1u) ? sc : 0.f;
            const uint2 lo4 = Op<T>::pack4(e[0], e[1], e[2], e[3]);
            const uint2 hi4 = Op<T>::pack4(e[4], e[5], e[6], e[7]);
            const uint4 idf = make_uint4(lo4.x, lo4.y, hi4.x, hi4.y);
#pragma unroll
            for (int row = 0; row < RPW; ++row) {
                if (c2)
                    acc[row][F][2 + ctl] = Op<T>::mfma16(idf, xf[row], acc[row][F][2 + ctl]);
                else
                    acc[row][F][ctl] = Op<T>::mfma16(idf, xf[row], acc[row][F][ctl]);
            }
        }
    };

    // the same for ONE of the wave's rows
    auto add_identity1 = [&](int row, const uint4& xf, float sc, int c2, auto freq_tag) {
        constexpr int F = decltype(freq_tag)::value;
#pragma unroll
        for (int ctl = 0; ctl < 2; ++ctl) {
            // (the empty asm keeps hipcc from building every identity fragment of the kernel once and holding them all - they are loop
            // invariant: 16 to 64 registers at the pressure peak)
            asm volatile("" : "+v"(sc));
            float e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = ((id_mask >> (8 * ctl + k)) & 1u) ? sc : 0.f;
            const uint2 lo4 = Op<T>::pack4(e[0], e[1], e[2], e[3]);
            const uint2 hi4 = Op<T>::pack4(e[4], e[5], e[6], e[7]);
            const uint4 idf = make_uint4(lo4.x, lo4.y, hi4.x, hi4.y);
            if (c2)
                acc[row][F][2 + ctl] = Op<T>::mfma16(idf, xf, acc[row][F][2 + ctl]);
            else
                acc[row][F][ctl] = Op<T>::mfma16(idf, xf, acc[row][F][ctl]);
        }
    };

    issue_w(0, 0);
    issue_act();

    int n = 0;
    for (int t = t_lo; t < t_hi; ++t) {
        int tty, ttx;
        tile_pos(t, tiles_x, tiles_y, &tty, &ttx);
        const int y0 = tty * TILE_H;
        const int x0 = ttx * TILE_W;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + 16 * ct + 4 * sl);
#pragma unroll
            for (int row = 0; row < RPW; ++row) {
                acc[row][0][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[row][1][ct] = bv;      // y(2j) and y(2j + 1) both take + m1
                acc[row][2][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[row][3][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }

        for (int c = 0; c < nch; ++c, ++n) {
            // item n has landed: each wave waits for its own DMAs, then the barrier.  A tile's first item after an epilogue does not wait
            // (vmcnt counts stores too; its DMAs were waited for ahead of the epilogue's stores).
            if (c > 0 || t == t_lo) FW_WAIT_VMCNT(0);
            __syncthreads();
            // ---- the wave's four halo rows -> 16 transformed fragments; the raw centre columns of its two output rows for x hi ------------
            uint4 V[4][4];
            const bool own = !STORE && c < 2 && p.in_id_scale != 0.f;   // the conv's own input channels [32c, 32c + 32) are a residual too
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint4* row = lds + (RPW * wave + r) * WN_ROWP;
                const uint4 d0 = row[rd[0]], d1 = row[rd[1]], d2 = row[rd[2]], d3 = row[rd[3]];
                V[r][0] = wn_sub(d0, d2);
                V[r][1] = wn_add(d1, d2);
                V[r][2] = wn_sub(d2, d1);
                V[r][3] = wn_sub(d1, d3);
                // x hi of output row r - 1 = the raw centre columns 2j (d1) and 2j + 1 (d2) of this halo row: added while they are in
                // registers (kept until after the barrier they were 16 more registers at the kernel's pressure peak)
                if ((r == 1 || r == 2) && own) {
                    add_identity1(r - 1, d1, p.in_id_scale, c, std::integral_constant<int, 0>{});
                    add_identity1(r - 1, d2, -p.in_id_scale, c, std::integral_constant<int, 3>{});
                }
            }
            __syncthreads();   // every wave holds its fragments: the stage is free for the next item's DMA
            const bool more = n + 1 < nitems;
            const int c1 = (c + 1 == nch) ? 0 : c + 1;
#ifndef FW_WN_DMA_K
#define FW_WN_DMA_K -1      // K-loop step behind which the next item's two DMA batches go out (-1: ahead of the loop)
#endif
            if (FW_WN_DMA_K < 0 && more) {
                issue_w(c1, (n + 1) & 1);
                issue_act();
            }
            // residual plane c (c < n_id): straight from HBM into B-fragment registers, both columns of the lane's pair.  The loads go out
            // in the LAST third of the K loop - two of the four halo rows' fragments are dead by then, which is where their 16 registers
            // come from - behind the item's DMA batches (hipcc's wait for them also sits out every LDS-DMA issued before them).  An item
            // without a plane reads the zero page (no branch inside the loop).
            uint4 ix0[RPW], ix1[RPW];
            const bool has_id = PLANES && c < p.n_id;
            // ---- 48 weight fragments, each feeding the wave's two rows ------------------------------------------------------------------
            const uint4* wl = lds + WN_ACT_REGION + (n & 1) * WN_W_REGION + lane;
#ifndef FW_WN_RING
#define FW_WN_RING 3
#endif
#ifndef FW_WN_RES_K
#define FW_WN_RES_K 17     // K-loop step behind which the residual plane's four loads go out: the first halo row's fragments are dead from step 16 on
#endif
            constexpr int RING = FW_WN_RING;
            uint4 wf[RING];
#pragma unroll
            for (int k = 0; k < RING - 1; ++k) wf[k] = wl[k * 64];
            FW_SB();
#pragma unroll
            for (int k = 0; k < WN_WFR; ++k) {
                const int dy = k / 16, f = (k / 4) & 3, ct = k & 3;
                if (k + RING - 1 < WN_WFR) wf[(k + RING - 1) % RING] = wl[(k + RING - 1) * 64];
                FW_SB();
                acc[0][f][ct] = Op<T>::mfma16(wf[k % RING], V[dy][f], acc[0][f][ct]);
                acc[1][f][ct] = Op<T>::mfma16(wf[k % RING], V[dy + 1][f], acc[1][f][ct]);
                FW_SB();
                if (k == FW_WN_DMA_K && more) {
                    issue_w(c1, (n + 1) & 1);
                    issue_act();
                    FW_SB();
                }
                if constexpr (PLANES) {
                    if (k == FW_WN_RES_K) {   // wave-uniform base + per-lane 32-bit offsets; a chunk without a plane re-reads plane 0 (its fragments are not used)
                        const char* plane = reinterpret_cast<const char*>(p.in) + p.chunk_off[has_id ? c : 0];
#pragma unroll
                        for (int row = 0; row < RPW; ++row) {
                            const int y = y0 + RPW * wave + row, x = x0 + 2 * j;
                            const int cy = y < p.H ? y : p.H - 1, cx0 = x < p.W ? x : p.W - 1, cx1 = x + 1 < p.W ? x + 1 : p.W - 1;
                            const unsigned rowb = (unsigned)cy * (unsigned)p.W;
                            ix0[row] = *reinterpret_cast<const uint4*>(plane + ((rowb + cx0) * (unsigned)p.in_cstride + 8 * sl) * 2u);
                            ix1[row] = *reinterpret_cast<const uint4*>(plane + ((rowb + cx1) * (unsigned)p.in_cstride + 8 * sl) * 2u);
                        }
                        FW_SB();
                    }
                }
            }
            if constexpr (PLANES) {
                if (has_id) {
#pragma unroll
                    for (int row = 0; row < RPW; ++row) {
                        const int y = y0 + RPW * wave + row, x = x0 + 2 * j;
                        if (!(y < p.H && x < p.W)) ix0[row] = make_uint4(0, 0, 0, 0);
                        if (!(y < p.H && x + 1 < p.W)) ix1[row] = make_uint4(0, 0, 0, 0);
                    }
                    add_identity(ix0, p.id_scale[c], c & 1, std::integral_constant<int, 0>{});
                    add_identity(ix1, -p.id_scale[c], c & 1, std::integral_constant<int, 3>{});
                }
            }
        }

        // ---- epilogue: the next tile's first item is in flight: wait for it here, ahead of the stores ---------------------------------------
        FW_WAIT_VMCNT(0);
        const int ls = (sl & 1) ? 2 + (sl >> 1) : (sl >> 1);
        const int npass = (!STORE && p.out_lo) ? 2 : 1;
#pragma unroll
        for (int row = 0; row < RPW; ++row) {
            const int y = y0 + RPW * wave + row;
            if (y >= p.H) continue;   // wave-uniform
            // output transform into frequencies 0 (column 2j) and 3 (column 2j + 1)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                f32x4 ya = acc[row][0][ct] + acc[row][1][ct] + acc[row][2][ct];
                f32x4 yb = acc[row][1][ct] - acc[row][2][ct] - acc[row][3][ct];
                if constexpr (STORE) {
                    if (p.act == 1) {
                        ya = lrelu4(ya);
                        yb = lrelu4(yb);
                    }
                } else {
                    ya = ya * p.s1;
                    yb = yb * p.s1;
                }
                acc[row][0][ct] = ya;
                acc[row][3][ct] = yb;
            }
            for (int pass = 0; pass < npass; ++pass) {
                char* lane_base = reinterpret_cast<char*>(pass ? p.out_lo : p.out) + ((long)(2 * j) * p.out_cstride + p.out_coff + ls * 8) * 2;
                const long rowoff = ((long)y * p.W + x0) * p.out_cstride * 2;
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        f32x4 oa = e ? acc[row][3][2 * c2] : acc[row][0][2 * c2], ob = e ? acc[row][3][2 * c2 + 1] : acc[row][0][2 * c2 + 1];
                        if (pass) {
                            oa = oa - Op<T>::unpack4(Op<T>::pack4(oa[0], oa[1], oa[2], oa[3]));
                            ob = ob - Op<T>::unpack4(Op<T>::pack4(ob[0], ob[1], ob[2], ob[3]));
                        }
                        const uint2 pa = Op<T>::pack4(oa[0], oa[1], oa[2], oa[3]);
                        const uint2 pb = Op<T>::pack4(ob[0], ob[1], ob[2], ob[3]);
                        const u32x2 sx = __builtin_amdgcn_permlane16_swap(pa.x, pb.x, false, false);
                        const u32x2 sy = __builtin_amdgcn_permlane16_swap(pa.y, pb.y, false, false);
                        if (x0 + 2 * j + e < p.W)
                            store16(lane_base + rowoff + (long)e * p.out_cstride * 2 + (long)c2 * p.out_pstride * 2, make_uint4(sx[0], sy[0], sx[1], sy[1]));
                    }
            }
        }
    }
}

#undef FW_SB

// Fragment order: [chunk c][tap row dy][frequency f][16-channel tile ct][lane][e], value = (G g)_f of
//   g = w[cout = 16 ct + (lane & 15)][cin = 32 c + 8 (lane >> 4) + e][dy][0..2]      (zero outside), rounded to the operand type once.
size_t pack_conv3x3_wino_weights(DType dt, const float* w, int cout, int cin, int cin_chunks, uint16_t* dst) {
    const size_t n = (size_t)cin_chunks * WN_WFR * 64 * 8;
    if (!dst) return n;
    size_t o = 0;
    for (int c = 0; c < cin_chunks; ++c)
        for (int dy = 0; dy < 3; ++dy)
            for (int f = 0; f < 4; ++f)
                for (int ct = 0; ct < 4; ++ct)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = 16 * ct + (lane & 15), ci = 32 * c + 8 * (lane >> 4) + e;
                            float u = 0.f;
                            if (co < cout && ci < cin) {
                                const float* g = w + ((size_t)co * cin + ci) * 9 + dy * 3;
                                u = f == 0 ? g[0] : f == 1 ? 0.5f * ((g[0] + g[1]) + g[2]) : f == 2 ? 0.5f * ((g[0] - g[1]) + g[2]) : g[2];
                            }
                            dst[o++] = f32_to_operand(dt, u);
                        }
    return n;
}

void launch_conv3x3_wino_split(const ConvParams& p_in, hipStream_t stream) {
    ConvParams p = p_in;
    p.zeros = conv_zero_page();
    if (p.H <= 0 || p.W <= 0 || p.cin_chunks <= 0 || p.upsample2x || !p.out || p.n_id < 0 || p.n_id > 6 || p.n_id > p.cin_chunks || p.in_cstride < 32 ||
        (p.in_cstride & 7) || (p.out_cstride & 7) || (p.out_coff & 7))
        throw Error(1, "conv3x3_wino: bad problem");
    const int tiles = ((p.W + TILE_W - 1) / TILE_W) * ((p.H + TILE_H - 1) / TILE_H);
    const int cus = conv_num_cus();
    dim3 grid(tiles < cus ? tiles : cus), block(64 * NWAVES);
    if (p.n_id > 0)
        hipLaunchKernelGGL((conv3x3_wino_split_kernel<true, false>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((conv3x3_wino_split_kernel<false, false>), grid, block, 0, stream, p);
    FW_HIP_CHECK(hipGetLastError());
}

// lrelu(conv3x3 + bias) (act = 1) or conv3x3 + bias, 64 output channels, typed planes out: the ConvParams of launch_conv3x3(dt, 2, EPI_STORE, ...)
void launch_conv3x3_wino_store(const ConvParams& p_in, hipStream_t stream) {
    ConvParams p = p_in;
    p.zeros = conv_zero_page();
    p.n_id = 0;
    p.out_lo = nullptr;
    if (p.H <= 0 || p.W <= 0 || p.cin_chunks <= 0 || p.upsample2x || !p.out || p.in_cstride < 32 || (p.in_cstride & 7) || (p.out_cstride & 7) || (p.out_coff & 7) ||
        p.out_f32 || p.act > 1)
        throw Error(1, "conv3x3_wino: bad problem");
    const int tiles = ((p.W + TILE_W - 1) / TILE_W) * ((p.H + TILE_H - 1) / TILE_H);
    const int cus = conv_num_cus();
    dim3 grid(tiles < cus ? tiles : cus), block(64 * NWAVES);
    hipLaunchKernelGGL((conv3x3_wino_split_kernel<false, true>), grid, block, 0, stream, p);
    FW_HIP_CHECK(hipGetLastError());
}

}  // namespace fw
