// conv3x3 (stride 1, zero pad 1) as an implicit GEMM on the CDNA4 matrix cores.
//
// This is kernel K1/K2/K3/K4 of SURVEY.md §8a: every 3x3 convolution of the RRDBNet that the
// reference runs through third-party basicsr (call site reference
// src/framewright/processors/pytorch_realesrgan.py:107-127,223) and whose residual-dense arithmetic is
// spelled out in-tree at src/framewright/processors/aesrgan_face.py:171-204 (ResidualDenseBlock / RRDB)
// and :249-269 (trunk + nearest-x2 upsample tail).
//
// GEMM orientation (chosen for NHWC stores, not translated from any CUDA tiling):
//     D[cout][pixel] += W[cout][k] * X[k][pixel]        k = (tap, cin)
//   * A operand  = weights, pre-packed on the host into 1-KiB v_mfma_f32_16x16x32 fragments (16 output channels x 32 input
//                  channels), staged through LDS by LDS-DMA one item ahead;
//   * B operand  = activations: a 16-pixel half row x 32 input channels, read from an XOR-swizzled LDS halo tile with
//                  ds_read_b128 (8 consecutive channels of one pixel = 16 bytes per lane);
//   * D          = the PIXEL is on the lane, 4 consecutive output channels sit in 4 consecutive registers; two tiles'
//                  fragments are paired with v_permlane16_swap into 16-byte NHWC stores straight from registers.
//
// Work decomposition: persistent 512-thread workgroups (8 waves, two per SIMD, 2 output rows each), one per CU, each
// walking a contiguous XCD-banded range of 16x32-pixel tiles as one flattened (tile, 32-channel chunk) pipeline; per
// item the 18x34-pixel halo tile (40 KiB) and the chunk's weight fragments arrive by batched LDS-DMA
// (global_load_lds_dwordx4).  conv_common.h holds the item (conv_item) and the DMA helpers; DESIGN.md section 6 the
// measurements behind every choice.
#include <mutex>
#include <cstdlib>
#include "fw_internal.h"
#include "conv_common.h"

namespace fw {

template <typename T, int CT, int EPI>
__global__ __launch_bounds__(64 * NWAVES, WAVES_PER_SIMD) void conv3x3_mfma_kernel(const ConvParams p_in) {
    ConvParams p = p_in;
    if constexpr (EPI == EPI_STORE || EPI == EPI_RESIDUAL) {
        if (p_in.n_groups > 1) {   // grid.y = output-channel group of a wide conv (scalar pointer arithmetic, once per workgroup)
            const int g = blockIdx.y;
            p.wpk = reinterpret_cast<const char*>(p_in.wpk) + (size_t)g * p_in.wpk_gstride;
            p.bias = p_in.bias + 32 * CT * g;
            if (p_in.chan_scale) p.chan_scale = p_in.chan_scale + 32 * CT * g;
            p.out_coff = p_in.out_coff + 32 * CT * g;
            p.f32_coff = p_in.f32_coff + 32 * CT * g;
            if (p_in.f32_native) {       // accumulator-native fp32 planes: one region per group
                const size_t gs = (size_t)g * p_in.f32_gstride;
                if (p_in.res1) p.res1 = p_in.res1 + gs;
                if (p_in.res2) p.res2 = p_in.res2 + gs;
                if (p_in.out_f32) p.out_f32 = p_in.out_f32 + gs;
            }
        }
    }
    using SM = Smem<CT>;
    __shared__ __attribute__((aligned(16))) uint4 lds[SM::TOTAL];
    constexpr int NA = SM::NA;
    constexpr int NW = 2 * CT;  // accumulator tiles of 16 output channels

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int q = lane & 15;   // pixel within a 16-pixel half row (B/D column), output channel within a tile (A row)
    const int sl = lane >> 4;  // 8-channel slot of the 32-channel chunk (A/B k index), 4-channel group of the D tile
    constexpr bool SPLIT = EPI == EPI_RESIDUAL_SPLIT;
    FW_STAMP_INIT();

    // ---- persistent workgroup: a contiguous range of tiles ---------------------------------------------------
    // Blocks b and b+8 share an XCD (and its L2): logical id lb puts the blocks of one XCD on a contiguous band of
    // tiles, so vertically neighbouring tiles (shared halo rows) and the weights are served by one L2.
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
    const int nitems = (t_hi - t_lo) * nch;  // (tile, chunk) pairs, walked as one pipelined sequence

    // ---- per-lane DMA plan ------------------------------------------------------------------------------------
    // Piece i of this lane = 16 bytes of halo position (row, px), slot s; a wave owns ACT_ITERS consecutive KiB of the stage
    // so that its pieces go out as one batch (one M0, conv_common.h).  The source of a piece is the wave-uniform address of
    // the tile's halo origin (row 0, px 0 = one pixel up-left of the tile) in the chunk's plane plus the loop-constant byte
    // offset; pieces outside the image read the zero page through a per-lane address (border tiles only).
    const int ups = p.upsample2x;
    const int Ws = ups ? (p.W >> 1) : p.W;
    unsigned relb[ACT_ITERS];  // byte offset from the halo origin, plus (4 - i) KiB: the batch's immediate takes that off again
    auto piece_pos = [&](int i, int* row, int* px) {
        const int idx = (ACT_ITERS * wave + i) * 64 + lane;
        const int rw = idx / ROW_PIECES;
        *px = (idx - rw * ROW_PIECES) >> 2;
        *row = (idx < ACT_PIECES) ? rw : -1;  // -1: the pad pieces behind the 18x34 image
    };
#pragma unroll
    for (int i = 0; i < ACT_ITERS; ++i) {
        const int idx = (ACT_ITERS * wave + i) * 64 + lane;
        const int row = idx / ROW_PIECES;
        const int rm = idx - row * ROW_PIECES;
        const int px = rm >> 2;
        const int s = (rm & 3) ^ halo_swz(px);  // which 8-channel slot lands at this LDS position
        // nearest-x2 input reads source pixel (y >> 1, x >> 1); ((row - 1) >> 1) + 1 >= 0 for row >= 0
        const int srow = ups ? (((row - 1) >> 1) + 1) : row;
        const int spx = ups ? (((px - 1) >> 1) + 1) : px;
        relb[i] = (unsigned)(((srow * Ws + spx) * p.in_cstride + s * 8) * 2) + (unsigned)(4 - i % 5) * 1024u;
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
    const char* in = reinterpret_cast<const char*>(p.in);
    const char* w_b = reinterpret_cast<const char*>(p.wpk);
    const unsigned lane16 = lane * 16;
    const long chunk_bytes = p.in_pstride * 2;

    // The activation DMA stream runs NA-1 items ahead of the compute and walks (tile, chunk) in order: a_t/a_c is the item
    // it issues next.
    unsigned a_ok = 0;             // bit i: piece i of this lane is inside the image
    bool a_all = true;             // uniform: every piece of every lane of this wave is
    const char* a_src = nullptr;   // uniform: halo origin of (tile a_t, chunk a_c)
    int a_n = 0, a_t = t_lo, a_c = 0;
    auto plan_tile = [&]() {
        int aty, atx;
        tile_pos(a_t, tiles_x, tiles_y, &aty, &atx);
        const int ty0 = aty * TILE_H, tx0 = atx * TILE_W;
        const int sy0 = ups ? (ty0 >> 1) : ty0;
        const int sx0 = ups ? (tx0 >> 1) : tx0;
        a_src = in + ((long)(sy0 - 1) * Ws + (sx0 - 1)) * p.in_cstride * 2;
        a_ok = 0;
#pragma unroll
        for (int i = 0; i < ACT_ITERS; ++i) {
            int row, px;
            piece_pos(i, &row, &px);
            const int gy = ty0 - 1 + row;
            const int gx = tx0 - 1 + px;
            if (row >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) a_ok |= 1u << i;
        }
        a_all = __builtin_amdgcn_readfirstlane(__all(a_ok == (1u << ACT_ITERS) - 1u)) != 0;
#ifdef FW_FORCE_SLOW_DMA  // timing experiment: every tile takes the per-piece border path
        a_all = false;
#endif
    };
    // the ACT_ITERS pieces of the next activation item
    auto issue_act = [&]() {
        if (a_c == 0) plan_tile();
        const unsigned dst = (unsigned)((a_n % NA) * ACT_REGION + ACT_ITERS * wave * 64);  // the wave's first piece
        if (a_all) {
#pragma unroll
            for (int bt = 0; bt < ACT_ITERS / 5; ++bt) glds16_batch_a(a_src, relb + 5 * bt, lds_base + (dst + (5 * bt + 4) * 64) * 16u);
        } else {
#pragma unroll
            for (int i = 0; i < ACT_ITERS; ++i)
                glds16_v(((a_ok >> i) & 1u) ? a_src + (relb[i] - (unsigned)(4 - i % 5) * 1024u) : reinterpret_cast<const char*>(p.zeros),
                         lds_base + (dst + i * 64) * 16u);
        }
        ++a_n;
        if (++a_c == nch) {
            a_c = 0;
            ++a_t;
        } else {
            a_src += chunk_bytes;
        }
    };
    // The weight fragments of chunk c into weight stage ws: the 18 fragments of each 32-output-channel half go to four waves
    // (5, 5, 5, 3), waves 0-3 the first half, waves 4-7 the second (CT == 2).  They are OLDER than the activation pieces
    // issued after them, which is what the counted vmcnt of the 3-stage pipeline relies on.
    auto issue_w = [&](int c, int ws) {
        const int half = wave / (NWAVES / 2), k = wave % (NWAVES / 2);
        if (half < CT)
            issue_w_half(w_b + (size_t)c * (W_FRAGS * CT * 1024) + W_FRAGS * half * 1024,
                         lds_base + (unsigned)(SM::W_BASE + ws * SM::W_REGION + W_FRAGS * half * 64) * 16u, k, lane16);
    };

    // ---- fragment-read plan ------------------------------------------------------------------------------------
    int rd_off[3][2];  // [dx][ph]: piece index of (halo row RPW*wave, px 16*ph + q + dx, slot sl) for this lane
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int px = 16 * ph + q + dx;
            rd_off[dx][ph] = (RPW * wave) * ROW_PIECES + px * 4 + (sl ^ halo_swz(px));
        }

    // identity A-fragments (SPLIT): tile ctl of a 32-channel residual plane, row = cout 16*ctl + q, k = 8*sl + j -> 1 where
    // cout == k, as a lane-private bit mask over (ctl, j); scaled per plane when used
    unsigned id_mask = 0;
    if constexpr (SPLIT) {
#pragma unroll
        for (int ctl = 0; ctl < 2; ++ctl)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (16 * ctl + q == 8 * sl + j) id_mask |= 1u << (8 * ctl + j);
    }

    f32x4 acc[RPW][NW][2];  // [row][16-channel tile][16-pixel half]: pixel 16*ph + q, channels 16*w + 4*sl + j

    // SPLIT: acc[row][2*c2 + ctl][ph] += sc * xf[row][ph] (32 channels of the wave's RPW x 32 pixels as B fragments): 8 MFMAs
    // with scaled identity A-fragments.  Products are exact, the sum is fp32.
    auto add_identity = [&](const uint4 (&xf)[RPW][2], float sc, int c2) {
#pragma unroll
        for (int ctl = 0; ctl < 2; ++ctl) {
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = ((id_mask >> (8 * ctl + j)) & 1u) ? sc : 0.f;
            const uint2 lo4 = Op<T>::pack4(e[0], e[1], e[2], e[3]);
            const uint2 hi4 = Op<T>::pack4(e[4], e[5], e[6], e[7]);
            const uint4 idf = make_uint4(lo4.x, lo4.y, hi4.x, hi4.y);
#pragma unroll
            for (int row = 0; row < RPW; ++row)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    if (c2)
                        acc[row][NW - 2 + ctl][ph] = Op<T>::mfma16(idf, xf[row][ph], acc[row][NW - 2 + ctl][ph]);
                    else
                        acc[row][ctl][ph] = Op<T>::mfma16(idf, xf[row][ph], acc[row][ctl][ph]);
                }
        }
    };

    // ---- pipeline prologue ------------------------------------------------------------------------------------
    // Issue order per boundary is [weights(n+1), activations(n+NA-1)]; vmcnt retires in order, so at boundary n
    // "all but the youngest ACT_ITERS DMAs" == everything up to and including weights(n) and activations(n).
    issue_w(0, 0);
    issue_act();
    if (NA == 3 && nitems > 1) issue_act();

    // timing ablations: build with -DFW_CONV_DEBUG=<bits> (2 = no LDS-DMA, 4 = no epilogue)
#ifndef FW_CONV_DEBUG
#define FW_CONV_DEBUG 0
#endif
    constexpr int dbg = FW_CONV_DEBUG;
    int n = 0;
    for (int t = t_lo; t < t_hi; ++t) {
        int tty, ttx;
        tile_pos(t, tiles_x, tiles_y, &tty, &ttx);
        const int y0 = tty * TILE_H;
        const int x0 = ttx * TILE_W;
        // accumulators start at the bias (re-read per tile: <= 256 B from L1/L2, cheaper than live registers)
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + 16 * w + 4 * sl);
#pragma unroll
            for (int row = 0; row < RPW; ++row)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) acc[row][w][ph] = bv;
        }
        FW_STAMP(4);  // tile setup

        for (int c = 0; c < nch; ++c, ++n) {
            // item n has landed (each wave waits for its own DMAs, then the barrier) and every wave is done reading
            // the stages that are refilled during this item
            // A tile's first item after an epilogue does not wait: vmcnt counts stores too (gfx9), so a wait behind the
            // epilogue's stores would sit out their round trip to memory; its DMAs were waited for before the epilogue.
            if (c > 0 || t == t_lo) {
                if (NA == 3 && n + 1 < nitems)
                    FW_WAIT_VMCNT(ACT_ITERS);  //: activations(n+1) may stay in flight
                else
                    FW_WAIT_VMCNT(0);
            }
            FW_STAMP(5);  // this wave's own DMAs
            __syncthreads();
            FW_STAMP(0);  // barrier
            const bool do_w = n + 1 < nitems && !(dbg & 2);
            const bool do_a = n + NA - 1 < nitems && !(dbg & 2);
            const int c1 = (c + 1 == nch) ? 0 : c + 1;
            // SPLIT: residual plane c (c < n_id) goes straight from HBM to B-fragment registers - lane (pixel 16*ph + q, slot sl) takes
            // 8 channels of its two rows' pixels - consumed after the item's MFMAs: no LDS, no halo.  The loads go out from slot
            // FW_RES_SLOT of the item, BEHIND its two DMA batches (slots FW_DMA_SLOT_W / _A): hipcc waits for its own loads with
            // vmcnt(0), which also sits out every LDS-DMA issued after them - issued at the top of the item (round 2) that wait
            // drained the prefetch of the next item at the end of every residual item (rdb3's conv5: 4 of 6 items; measured in round 3,
            // DESIGN.md section 6.3); issued last, the DMAs in front of them have had the item's MFMAs to land.
            uint4 idx[RPW][2];
            const bool has_id = SPLIT && c < p.n_id;
            auto load_residual = [&]() {
                if constexpr (SPLIT) {
                    if (has_id) {
                        const char* plane = reinterpret_cast<const char*>(p.in) + p.chunk_off[c];
#pragma unroll
                        for (int row = 0; row < RPW; ++row)
#pragma unroll
                            for (int ph = 0; ph < 2; ++ph) {
                                const int y = y0 + RPW * wave + row, x = x0 + 16 * ph + q;
                                {
                                    // outside the image: the zero page (a select on the loaded VALUE would wait for the load here)
                                    const char* px = (y < p.H && x < p.W) ? plane + (((size_t)y * p.W + x) * p.in_cstride + 8 * sl) * 2
                                                                          : reinterpret_cast<const char*>(p.zeros);
                                    idx[row][ph] = *reinterpret_cast<const uint4*>(px);
                                }
                            }
                    }
                }
            };
            // The two DMA batches of this boundary (weights of item n+1, activations of item n+NA-1) go out in the shadow
            // of the first MFMAs: slot d of 36, compile-time after unrolling.
#ifndef FW_DMA_SLOT_W
#define FW_DMA_SLOT_W 2
#define FW_DMA_SLOT_A 6
#endif
#ifndef FW_DMA_SLOT_W2   // the younger half of the workgroup (waves NWAVES/2..): same slots unless told otherwise
#define FW_DMA_SLOT_W2 FW_DMA_SLOT_W
#define FW_DMA_SLOT_A2 FW_DMA_SLOT_A
#endif
            const bool young = wave >= NWAVES / 2;
#ifndef FW_RES_SLOT
#define FW_RES_SLOT 20
#endif
            auto dma_slot = [&](int d) {
                if (d == (young ? FW_DMA_SLOT_W2 : FW_DMA_SLOT_W)) {
                    if (do_w) issue_w(c1, (n + 1) & 1);
                } else if (d == (young ? FW_DMA_SLOT_A2 : FW_DMA_SLOT_A)) {
                    if (do_a) issue_act();
                } else if (d == FW_RES_SLOT) {
                    load_residual();
                }
            };

            const uint4* a = lds + (n % NA) * ACT_REGION;
            const uint4* wl = lds + SM::W_BASE + (n & 1) * SM::W_REGION + lane;
            // SPLIT: residual plane c (c < n_id) goes straight from HBM to B-fragment registers - lane (pixel 16*ph + q,
            // slot sl) takes 8 channels of its two rows' pixels - issued now, consumed after the item's MFMAs: no LDS, no
            // halo, and the latency hides under the item.
            auto& slot_fn = dma_slot;
            conv_item<T, NW, 0>(
                acc, a, wl, rd_off, [](int tap, int w) { return tap * NW + w; }, slot_fn,
                [&](const uint4 (&xc)[RPW][2]) {
                    if constexpr (SPLIT) {
                        // the conv's own input channels [32c, 32c+32) are a residual too: centre tap of the tile in LDS
                        if (c < CT && p.in_id_scale != 0.f) add_identity(xc, p.in_id_scale, c);
                    }
                },
                [](int) {});
            FW_STAMP(1);  // item compute
            if constexpr (SPLIT) {
                if (has_id) {
                    add_identity(idx, p.id_scale[c], c & 1);
                }
            }
            FW_STAMP(2);  // residual plane: wait for its loads + 8 identity MFMAs
        }

        // ---- epilogue (the DMA stream is already fetching the next tile) -----------------------------------------
        // the next tile's first item (n) has been in flight for an item: wait for it here, ahead of the epilogue's stores
        if (NA == 3 && n + 1 < nitems)
            FW_WAIT_VMCNT(ACT_ITERS);
        else
            FW_WAIT_VMCNT(0);
        FW_STAMP(5);
        if (dbg & 4) continue;
        if constexpr (EPI == EPI_IMAGE) {
            // output channels 0..2 = R,G,B: tile 0, registers j = 0..2 of the lanes with sl == 0
#pragma unroll
            for (int row = 0; row < RPW; ++row)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const int y = y0 + RPW * wave + row;
                    const int x = x0 + 16 * ph + q;
                    if (sl == 0 && y < p.img_H && x < p.img_W && y < p.H && x < p.W) {
                        const size_t pix = (size_t)y * p.img_W + x;
                        const float cr = acc[row][0][ph][0], cg = acc[row][0][ph][1], cb = acc[row][0][ph][2];
                        if (p.out_rgb) {
                            float* o = p.out_rgb + pix * 3;
                            o[0] = cr;
                            o[1] = cg;
                            o[2] = cb;
                        }
                        if (p.out_u16) {
                            uint16_t* o = p.out_u16 + pix * 3;
                            o[0] = (uint16_t)rintf(fminf(fmaxf(cb, 0.f), 1.f) * 65535.f);
                            o[1] = (uint16_t)rintf(fminf(fmaxf(cg, 0.f), 1.f) * 65535.f);
                            o[2] = (uint16_t)rintf(fminf(fmaxf(cr, 0.f), 1.f) * 65535.f);
                        }
                        if (p.out_u8) {
                            uint8_t* o = p.out_u8 + pix * 3;
                            o[0] = (uint8_t)rintf(fminf(fmaxf(cb, 0.f), 1.f) * 255.f);
                            o[1] = (uint8_t)rintf(fminf(fmaxf(cg, 0.f), 1.f) * 255.f);
                            o[2] = (uint8_t)rintf(fminf(fmaxf(cr, 0.f), 1.f) * 255.f);
                        }
                    }
                }
        } else {
            constexpr int NC = 32 * CT;
            if constexpr (SPLIT) {
#pragma unroll
                for (int row = 0; row < RPW; ++row)
#pragma unroll
                    for (int w = 0; w < NW; ++w)
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph) {
                            f32x4 o = acc[row][w][ph] * p.s1;
                            if (p.post_act) {   // IFNet's ResConv on a split trunk: LeakyReLU(0.2) after the residual add
#pragma unroll
                                for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.2f * o[j]);
                            }
                            acc[row][w][ph] = o;
                        }
            } else {
                // (1) fp32 side: residuals in, trunk out.  Native layout = the accumulator fragment order
                //     [tile][wave][row][w][ph][lane][4], one contiguous KiB per wave-instruction, no transposition;
                //     NHWC (op-level API) is the slow general form.
#pragma unroll
                for (int row = 0; row < RPW; ++row)
#pragma unroll
                    for (int w = 0; w < NW; ++w)
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph) {
                            const int y = y0 + RPW * wave + row;
                            const int x = x0 + 16 * ph + q;
                            const bool inside = y < p.H && x < p.W;
                            const size_t pix = (size_t)y * p.W + x;
                            const size_t nat = ((((((size_t)t * NWAVES + wave) * RPW + row) * NW + w) * 2 + ph) * 64 + lane) * 4;
                            const size_t lin = pix * (p.f32_cstride ? p.f32_cstride : NC) + p.f32_coff + 16 * w + 4 * sl;
                            const size_t fo = p.f32_native ? nat : lin;
                            const bool fok = p.f32_native || inside;
                            f32x4 o = acc[row][w][ph];
                            if constexpr (EPI == EPI_RESIDUAL) {
                                if (fok) {
                                    if (p.chan_scale) o = o * *reinterpret_cast<const f32x4*>(p.chan_scale + 16 * w + 4 * sl);
                                    o = o * p.s1 + *reinterpret_cast<const f32x4*>(p.res1 + fo);
                                    if (p.res2) o = o * p.s2 + *reinterpret_cast<const f32x4*>(p.res2 + fo);
                                    if (p.post_act) {
#pragma unroll
                                        for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.2f * o[j]);
                                    }
                                }
                            } else {
                                if (p.act == 1) {
                                    o = lrelu4(o);
                                } else if (p.act == 2) {  // PReLU, per-channel slopes in chan_scale (SRVGGNetCompact)
                                    const f32x4 sl4 = *reinterpret_cast<const f32x4*>(p.chan_scale + 16 * w + 4 * sl);
#pragma unroll
                                    for (int j = 0; j < 4; ++j) o[j] = o[j] > 0.f ? o[j] : sl4[j] * o[j];
                                }
                            }
                            if (p.out_f32 && fok) *reinterpret_cast<f32x4*>(p.out_f32 + fo) = o;
                            acc[row][w][ph] = o;
                        }
            }
            FW_STAMP(3);  // fp32-side epilogue
            // (2) typed NHWC output, 16 B per lane straight from the accumulators: v_permlane16_swap pairs the 8-byte fragments
            //     of two neighbouring 16-channel tiles so that a lane ends up with one whole 8-channel slot of its pixel
            //     (ls; conv3x3_pair.hip has the lane map) and four lanes store a pixel's 64 contiguous bytes.  No transpose
            //     through LDS and no barrier: the older wave of a SIMD stores while the younger one still runs MFMAs.
            if (p.out) {
                // pass 0: the typed output; pass 1 (64-channel convs feeding a split trunk): lo = T(y - T(y)) into out_lo
                const int npass = (CT == 2 && p.out_lo) ? 2 : 1;
                const int ls = (sl & 1) ? 2 + (sl >> 1) : (sl >> 1);
                for (int pass = 0; pass < npass; ++pass) {
                    // store address = per-lane base (once per pass) + wave-uniform row / half-row / plane offsets (scalar)
                    char* lane_base = reinterpret_cast<char*>(pass ? p.out_lo : p.out) +
                                      ((long)q * p.out_cstride + p.out_coff + ls * 8) * 2;
#pragma unroll
                    for (int row = 0; row < RPW; ++row) {
                        const int y = y0 + RPW * wave + row;
                        if (y >= p.H) continue;  // wave-uniform
                        const long rowoff = ((long)y * p.W + x0) * p.out_cstride * 2;
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                            for (int c2 = 0; c2 < CT; ++c2) {
                                f32x4 oa = acc[row][2 * c2][ph], ob = acc[row][2 * c2 + 1][ph];
                                if (pass) {
                                    oa = oa - Op<T>::unpack4(Op<T>::pack4(oa[0], oa[1], oa[2], oa[3]));
                                    ob = ob - Op<T>::unpack4(Op<T>::pack4(ob[0], ob[1], ob[2], ob[3]));
                                }
                                const uint2 pa = Op<T>::pack4(oa[0], oa[1], oa[2], oa[3]);
                                const uint2 pb = Op<T>::pack4(ob[0], ob[1], ob[2], ob[3]);
                                const u32x2 sx = __builtin_amdgcn_permlane16_swap(pa.x, pb.x, false, false);
                                const u32x2 sy = __builtin_amdgcn_permlane16_swap(pa.y, pb.y, false, false);
                                if (x0 + 16 * ph + q < p.W)
                                    store16(lane_base + rowoff + (long)(16 * ph) * p.out_cstride * 2 + (long)c2 * p.out_pstride * 2,
                                            make_uint4(sx[0], sy[0], sx[1], sy[1]));
                            }
                    }
                }
            }
            FW_STAMP(6);  // typed store
        }
    }
    if constexpr (CT == 2 && (EPI == EPI_RESIDUAL || SPLIT)) FW_STAMP_FLUSH(p.stamps);
}

#undef FW_SB

#ifdef FW_PAIR_STAMP
unsigned long long* stamp_buffer(int which) {
    static unsigned long long* buf[2] = {nullptr, nullptr};
    if (!buf[which]) {
        FW_HIP_CHECK(hipMalloc((void**)&buf[which], 1024));
        FW_HIP_CHECK(hipMemset(buf[which], 0, 1024));
    }
    return buf[which];
}
extern "C" int fw_debug_stamps(int which, unsigned long long* out) {
    if (which < 0 || which > 1) return 1;
    if (hipMemcpy(out, stamp_buffer(which), 1024, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    (void)hipMemset(stamp_buffer(which), 0, 1024);
    return 0;
}
#endif

static_assert(ACT_ITERS % 5 == 0, "activation pieces go out in batches of five");

// 256 bytes of zeros per device: the DMA source of halo positions outside the image.
static const void* zero_page() {
    static std::mutex mu;
    static void* pages[64] = {};
    int dev = 0;
    FW_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) throw Error(1, "conv3x3: device ordinal out of range");
    std::lock_guard<std::mutex> lk(mu);
    if (!pages[dev]) {
        void* z = nullptr;
        FW_HIP_CHECK(hipMalloc(&z, 256));
        FW_HIP_CHECK(hipMemset(z, 0, 256));
        FW_HIP_CHECK(hipDeviceSynchronize());
        pages[dev] = z;
    }
    return pages[dev];
}

size_t f32_native_elems(int H, int W, int cout_tiles) {
    const size_t tiles = (size_t)((W + TILE_W - 1) / TILE_W) * ((H + TILE_H - 1) / TILE_H);
    return tiles * (TILE_H * TILE_W) * 32 * cout_tiles;
}

const void* conv_zero_page() { return zero_page(); }

static int num_cus() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        // FW_CONV_GRID: persistent workgroups per launch (default one per CU); half of them lets two frames run side by side on
        // two streams, each kernel on its own half of the CUs
        if (const char* e = getenv("FW_CONV_GRID")) {
            const int g = atoi(e);
            if (g > 0 && g < v) v = g;
        }
        return v;
    }();
    return n;
}

int conv_num_cus() { return num_cus(); }

template <typename T>
static void launch_typed(int cout_tiles, ConvEpilogue epi, const ConvParams& p, hipStream_t stream) {
    const int tiles = ((p.W + TILE_W - 1) / TILE_W) * ((p.H + TILE_H - 1) / TILE_H);
    // persistent workgroups: one per CU (LDS-limited), each walks a contiguous range of tiles
    const int ng = p.n_groups > 1 ? p.n_groups : 1;
    const int per_group = num_cus() / ng > 0 ? num_cus() / ng : 1;
    dim3 grid(tiles < per_group ? tiles : per_group, ng), block(64 * NWAVES);
    if (cout_tiles == 1 && epi == EPI_STORE)
        hipLaunchKernelGGL((conv3x3_mfma_kernel<T, 1, EPI_STORE>), grid, block, 0, stream, p);
    else if (cout_tiles == 2 && epi == EPI_STORE)
        hipLaunchKernelGGL((conv3x3_mfma_kernel<T, 2, EPI_STORE>), grid, block, 0, stream, p);
    else if (cout_tiles == 2 && epi == EPI_RESIDUAL)
        hipLaunchKernelGGL((conv3x3_mfma_kernel<T, 2, EPI_RESIDUAL>), grid, block, 0, stream, p);
    else if (cout_tiles == 1 && epi == EPI_RESIDUAL)   // 32-channel groups of a wide conv on a small map (IFNet's low-resolution blocks)
        hipLaunchKernelGGL((conv3x3_mfma_kernel<T, 1, EPI_RESIDUAL>), grid, block, 0, stream, p);
    else if (cout_tiles == 2 && epi == EPI_RESIDUAL_SPLIT)
        hipLaunchKernelGGL((conv3x3_mfma_kernel<T, 2, EPI_RESIDUAL_SPLIT>), grid, block, 0, stream, p);
    else if (cout_tiles == 1 && epi == EPI_IMAGE)
        hipLaunchKernelGGL((conv3x3_mfma_kernel<T, 1, EPI_IMAGE>), grid, block, 0, stream, p);
    else
        throw Error(1, "conv3x3: unsupported (cout_tiles, epilogue) combination");
    FW_HIP_CHECK(hipGetLastError());
}

void launch_conv3x3(DType dt, int cout_tiles, ConvEpilogue epi, const ConvParams& p_in, hipStream_t stream) {
    ConvParams p = p_in;
    p.zeros = zero_page();
#ifdef FW_PAIR_STAMP
    p.stamps = stamp_buffer(1);
#endif


    if (p.H <= 0 || p.W <= 0 || p.cin_chunks <= 0) throw Error(1, "conv3x3: empty problem");
    if (p.upsample2x && ((p.H | p.W) & 1)) throw Error(1, "conv3x3: upsample2x needs even output size");
    if (p.in_cstride < 32 || (p.in_cstride & 7) || p.in_pstride < 32 || (p.in_pstride & 7) ||
        (p.in_pstride == 32 && p.in_cstride < 32 * p.cin_chunks &&
         (long)(p.upsample2x ? p.H / 2 : p.H) * (p.upsample2x ? p.W / 2 : p.W) > 1))  // (a 1-pixel planar input IS interleaved)
        throw Error(1, "conv3x3: bad input channel/plane stride");
    if (p.out && cout_tiles == 2 && (p.out_pstride < 32 || (p.out_pstride & 7)))
        throw Error(1, "conv3x3: bad output plane stride");
    if (p.out && ((p.out_cstride & 7) || (p.out_coff & 7))) throw Error(1, "conv3x3: output slice must be 16-byte aligned");
    if (p.act == 2 && !p.chan_scale) throw Error(1, "conv3x3: PReLU needs its slopes");
    if (p.n_groups > 1 && ((epi != EPI_STORE && epi != EPI_RESIDUAL) || p.n_groups > 64 || p.wpk_gstride <= 0 || (p.wpk_gstride & 15) || (p.f32_native && p.f32_gstride < (long)f32_native_elems(p.H, p.W, cout_tiles))))
        throw Error(1, "conv3x3: output-channel groups need EPI_STORE / EPI_RESIDUAL and a 16-byte weight stride");
    if (p.out_lo && (cout_tiles != 2 || !p.out || epi == EPI_IMAGE)) throw Error(1, "conv3x3: out_lo needs a 64-channel typed output");
    if (epi == EPI_RESIDUAL_SPLIT) {
        if (cout_tiles != 2 || p.upsample2x || !p.out || p.n_id < 0 || p.n_id > 6 || p.n_id > p.cin_chunks)
            throw Error(1, "conv3x3: bad split-trunk residual problem");
        for (int i = 0; i < p.n_id; ++i)
            if (operand_to_f32(dt, f32_to_operand(dt, p.id_scale[i])) != p.id_scale[i])
                throw Error(1, "conv3x3: identity scale not representable in the operand type");
    }
    if (dt == DT_BF16)
        launch_typed<__bf16>(cout_tiles, epi, p, stream);
    else
        launch_typed<_Float16>(cout_tiles, epi, p, stream);
}

// ---- host-side packing ----------------------------------------------------------------------------------
uint16_t f32_to_operand(DType dt, float f) {
    if (dt == DT_BF16) {
        uint32_t u;
        memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
        u += 0x7fffu + ((u >> 16) & 1u);                                            // round to nearest even
        return (uint16_t)(u >> 16);
    }
    _Float16 hv = (_Float16)f;
    uint16_t o;
    memcpy(&o, &hv, 2);
    return o;
}

float operand_to_f32(DType dt, uint16_t v) {
    if (dt == DT_BF16) {
        uint32_t u = (uint32_t)v << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
    _Float16 hv;
    memcpy(&hv, &v, 2);
    return (float)hv;
}

// Fragment order: [chunk c][tap t = ky*3+kx][16-channel cout tile w][lane][j], value =
//   w[cout = 16*w + (lane & 15)][cin = 32*c + 8*(lane >> 4) + j][ky][kx]      (zero outside)
// which is exactly the v_mfma_f32_16x16x32 A-operand map (row = lane & 15, k = 8*(lane >> 4) + j).
size_t pack_conv3x3_weights(DType dt, const float* w, int cout, int cin, int cout_tiles, int cin_chunks,
                            uint16_t* dst) {
    const size_t n = (size_t)cin_chunks * 9 * 2 * cout_tiles * 64 * 8;
    if (!dst) return n;
    size_t o = 0;
    for (int c = 0; c < cin_chunks; ++c)
        for (int t = 0; t < 9; ++t)
            for (int wt = 0; wt < 2 * cout_tiles; ++wt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = 16 * wt + (lane & 15);
                        const int ci = 32 * c + 8 * (lane >> 4) + j;
                        float val = 0.f;
                        if (co < cout && ci < cin) val = w[((size_t)co * cin + ci) * 9 + t];
                        dst[o++] = f32_to_operand(dt, val);
                    }
    return n;
}

}  // namespace fw
