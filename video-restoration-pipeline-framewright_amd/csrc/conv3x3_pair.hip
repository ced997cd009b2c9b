// Two chained 3x3 convolutions of a residual dense block in ONE kernel:  x_a = lrelu(conv_a([x..]))  and
// x_b = lrelu(conv_b([x.., x_a])) — conv1+conv2 and conv3+conv4 of reference
// src/framewright/processors/aesrgan_face.py:184-187.
//
// Why: the unfused path is HBM-bound and conv_b re-reads everything conv_a just read.  Here every input
// chunk is staged ONCE and feeds BOTH convolutions while it sits in LDS (two accumulator sets, like a 64-output-channel
// conv whose second half belongs to conv_b); after the last shared chunk x_a is finished, written to LDS in the same
// swizzled halo-tile format the LDS-DMA produces, and one more K-step adds W_b[last chunk] * x_a.
//
// Geometry: both convs are evaluated on the same 16x32 "compute region" with the standard 18x34 halo tile.
// conv_b's zero-padded stencil over x_a is only valid one pixel inside the region, so a tile delivers 14x30 pixels
// (neighbouring tiles recompute the 1-pixel ring: +22 % MFMA work on these four convs, +12 % on the block, for about
// half of their HBM reads).  x_a outside the image is zero (= conv_b's zero padding).  Outputs leave the accumulators as
// 16-byte stores (v_permlane16_swap pairs the fragments of two 16-channel tiles), before the barrier.
// This is the RING form, used for small frames; conv3x3_pair_slide.hip removes the vertical ring and is the default from
// about 720p up (launch_conv3x3_pair below chooses).
#include <mutex>
#include <type_traits>
#include <cstdlib>
#include "fw_internal.h"
#include "conv_common.h"

namespace fw {

// timing ablations: build with -DFW_PAIR_DBG=<bits> (1 no activation DMA, 2 no weight DMA, 4 no emit, 8 no MFMA)
#ifndef FW_PAIR_DBG
#define FW_PAIR_DBG 0
#endif

#ifdef FW_PAIR_TH_ABLATION  // timing-only ablation (wrong pixels on the ring): what a tile step of 16 rows would cost
constexpr int PAIR_TH = FW_PAIR_TH_ABLATION;
#else
constexpr int PAIR_TH = TILE_H - 2;  // 14 valid rows per tile
#endif
constexpr int PAIR_TW = TILE_W - 2;  // 30 valid pixels per tile row

struct PairSmem {
    static constexpr int NA = 2;                          // activation stages (the x_a tile borrows the idle one)
    static constexpr int W_REGION = 2 * W_FRAGS * 64;     // weight chunk of conv_a (18 fragments) + conv_b (18)
    static constexpr int W_BASE = NA * ACT_REGION;
    static constexpr int W_ITERS = (2 * W_FRAGS + NWAVES - 1) / NWAVES;  // 5
    static constexpr int TOTAL = W_BASE + 2 * W_REGION;   // 152 KiB
};

template <typename T>
__global__ __launch_bounds__(64 * NWAVES, WAVES_PER_SIMD) void conv3x3_pair_kernel(const ConvPairParams p) {
    using SM = PairSmem;
    __shared__ __attribute__((aligned(16))) uint4 lds[SM::TOTAL];
    constexpr int NW = 4;  // accumulator tiles of 16 output channels: 0,1 = conv_a, 2,3 = conv_b

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int q = lane & 15;   // pixel within a 16-pixel half row (B/D column), output channel within a tile (A row)
    const int sl = lane >> 4;  // 8-channel slot of the chunk (A/B k index), 4-channel group of the D tile
    FW_STAMP_INIT();

    const int NB = gridDim.x;
    const int xcd = blockIdx.x & 7;
    const int qn = NB >> 3, rn = NB & 7;
    const int lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int tiles_x = (p.W + PAIR_TW - 1) / PAIR_TW;
    const int tiles_y = (p.H + PAIR_TH - 1) / PAIR_TH;
    const int ntiles = tiles_x * tiles_y;
    const int t_lo = (int)((long)lb * ntiles / NB);
    const int t_hi = (int)((long)(lb + 1) * ntiles / NB);
    if (t_lo >= t_hi) return;
    const int na = p.na;             // input chunks shared by both convs
    const int ipt = na + 1;          // items per tile: na shared chunks (both convs), then conv_b's x_a chunk
    const int nitems = (t_hi - t_lo) * ipt;

    // ---- per-lane DMA plan (as in conv3x3_mfma.hip; the compute region plays the role of the tile) ----------------------
    unsigned relb[ACT_ITERS];  // byte offset of this lane's piece from the region's halo origin, plus (4 - i) KiB (batch immediate)
    auto piece_pos = [&](int i, int* row, int* px) {
        const int idx = (ACT_ITERS * wave + i) * 64 + lane;
        const int rw = idx / ROW_PIECES;
        *px = (idx - rw * ROW_PIECES) >> 2;
        *row = (idx < ACT_PIECES) ? rw : -1;
    };
#pragma unroll
    for (int i = 0; i < ACT_ITERS; ++i) {
        const int idx = (ACT_ITERS * wave + i) * 64 + lane;
        const int row = idx / ROW_PIECES;
        const int rm = idx - row * ROW_PIECES;
        const int px = rm >> 2;
        const int s = (rm & 3) ^ halo_swz(px);
        relb[i] = (unsigned)(((row * p.W + px) * p.in_cstride + s * 8) * 2) + (unsigned)(4 - i % 5) * 1024u;
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
    const char* in = reinterpret_cast<const char*>(p.in);
    const char* wa_b = reinterpret_cast<const char*>(p.wpk_a);
    const char* wb_b = reinterpret_cast<const char*>(p.wpk_b);
    const unsigned lane16 = lane * 16;
    const long chunk_bytes = p.in_pstride * 2;

    // compute-region origin of tile t (may be -1: the region starts one pixel outside the tile's valid outputs)
    auto origin = [&](int t, int* oy, int* ox) {
        int ty, tx;
        tile_pos(t, tiles_x, tiles_y, &ty, &tx);
        *oy = ty * PAIR_TH - 1;
        *ox = tx * PAIR_TW - 1;
    };

    // activation fetch stream: per tile the chunks 0..na-1, each fetched once, as one batch of ACT_ITERS pieces per wave.
    // Pieces outside the image read the zero page through a per-lane address (border tiles only).
    unsigned f_ok = 0;
    bool f_all = true;             // uniform
    const char* f_src = nullptr;   // uniform: halo origin of (tile f_t, chunk f_c)
    int f_t = t_lo, f_c = 0;
    auto plan_tile = [&]() {
        int oy, ox;
        origin(f_t, &oy, &ox);
        f_src = in + ((long)(oy - 1) * p.W + (ox - 1)) * p.in_cstride * 2;
        f_ok = 0;
#pragma unroll
        for (int i = 0; i < ACT_ITERS; ++i) {
            int row, px;
            piece_pos(i, &row, &px);
            const int gy = oy - 1 + row;
            const int gx = ox - 1 + px;
            if (row >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) f_ok |= 1u << i;
        }
        f_all = __builtin_amdgcn_readfirstlane(__all(f_ok == (1u << ACT_ITERS) - 1u)) != 0;
    };
    auto issue_act = [&](int stage) {
        if (f_c == 0) plan_tile();
        const unsigned dst = (unsigned)(stage * ACT_REGION + ACT_ITERS * wave * 64);  // the wave's first piece
        if (f_all) {
#pragma unroll
            for (int bt = 0; bt < ACT_ITERS / 5; ++bt) glds16_batch_a(f_src, relb + 5 * bt, lds_base + (dst + (5 * bt + 4) * 64) * 16u);
        } else {
#pragma unroll
            for (int i = 0; i < ACT_ITERS; ++i)
                glds16_v(((f_ok >> i) & 1u) ? f_src + (relb[i] - (unsigned)(4 - i % 5) * 1024u) : reinterpret_cast<const char*>(p.zeros),
                         lds_base + (dst + i * 64) * 16u);
        }
        if (++f_c == na) {
            f_c = 0;
            ++f_t;
        } else {
            f_src += chunk_bytes;
        }
    };
    // weights of item j of a tile -> weight stage ws: fragments [0,18) = conv_a chunk j (waves 0-3: 5, 5, 5, 3 fragments),
    // [18,36) = conv_b chunk j (waves 4-7); the last item (j == na) only has conv_b's chunk
    auto issue_w = [&](int j, int ws) {
        const int half = wave / (NWAVES / 2), k = wave % (NWAVES / 2);
        if (half == 1 || j < na)
            issue_w_half((half ? wb_b : wa_b) + (size_t)j * (W_FRAGS * 1024),
                         lds_base + (unsigned)(SM::W_BASE + ws * SM::W_REGION + W_FRAGS * half * 64) * 16u, k, lane16);
    };

    int rd_off[3][2];  // [dx][ph]: piece index of (halo row RPW*wave, px 16*ph + q + dx, slot sl)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int px = 16 * ph + q + dx;
            rd_off[dx][ph] = (RPW * wave) * ROW_PIECES + px * 4 + (sl ^ halo_swz(px));
        }
    // weight fragment of (tap, tile w) inside a weight stage
    auto widx = [](int tap, int w) { return w < 2 ? tap * 2 + w : W_FRAGS + tap * 2 + (w - 2); };

    f32x4 acc[RPW][NW][2];  // [row][tile][16-pixel half]: pixel 16*ph + q, channels 16*(w & 1) + 4*sl + j of conv w >> 1

    // ---- emit: accumulators -> LeakyReLU -> typed, without a transpose through LDS and BEFORE the barrier ------------------
    // A lane's D fragments hold channels 4*sl.. of tile w0 and of tile w0+1 for pixel 16*ph + q (8 B each).  One
    // v_permlane16_swap per dword (rows of 16 lanes: odd rows of the first operand <-> even rows of the second) leaves every
    // lane with 16 contiguous bytes: lanes with even sl get channels 4*sl..4*sl+7 of tile w0 (their own 8 B + those of lane
    // ^ 16), lanes with odd sl channels 4*(sl-1)..+7 of tile w0+1.  That is one whole 8-channel slot of the pixel:
    //     ls = sl even ? sl / 2 : 2 + sl / 2
    // so the global store is 16 B per lane / 64 contiguous bytes per pixel straight from registers, and x_a goes into its LDS
    // tile with one ds_write_b128 per fragment.  Everything up to the LDS write needs no barrier: a wave converts and stores
    // as soon as ITS last item is done - the older wave of a SIMD while the younger one still runs MFMAs, the younger one
    // with the VALU to itself (with the convert behind the barrier both waves of a SIMD converted at the same time: 18 % of
    // the kernel, phase stamps in DESIGN.md section 6).
    const int ls = (sl & 1) ? 2 + (sl >> 1) : (sl >> 1);
    const unsigned st_off = (unsigned)((q * p.out_cstride + ls * 8) * 2);  // store offset from (row, compute-region px 16*ph)
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
    // valid interior of the compute region -> global plane (pixels 0 / 31 and rows 0 / 15 are the recompute ring)
    auto store_out = [&](const uint4 (&pk)[RPW][2], int oy, int ox, T* plane, auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int row = 0; row < RPW; ++row) {
            const int cr = RPW * wave + row;
            const int gy = oy + cr;
            const bool row_ok = cr >= 1 && cr <= PAIR_TH && (INTERIOR || (unsigned)gy < (unsigned)p.H);  // wave-uniform
            if (!row_ok) continue;
            char* rowbase = reinterpret_cast<char*>(plane) + ((long)gy * p.W + ox) * p.out_cstride * 2;  // wave-uniform
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int cp = 16 * ph + q;
                if (cp >= 1 && cp <= PAIR_TW && (INTERIOR || (unsigned)(ox + cp) < (unsigned)p.W))
                    store16(rowbase + (long)(16 * ph) * p.out_cstride * 2 + st_off, pk[row][ph]);
            }
        }
    };
    // x_a -> its halo-format tile in LDS (zero outside the image = conv_b's zero padding; border tiles only)
    auto write_xa = [&](const uint4 (&pk)[RPW][2], uint4* xa, int oy, int ox, auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int row = 0; row < RPW; ++row) {
            const int cr = RPW * wave + row;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int cp = 16 * ph + q, hp = cp + 1;
                uint4 v = pk[row][ph];
                if (!INTERIOR && !((unsigned)(oy + cr) < (unsigned)p.H && (unsigned)(ox + cp) < (unsigned)p.W)) v = make_uint4(0, 0, 0, 0);
                xa[((cr + 1) * HALO_W + hp) * 4 + (ls ^ halo_swz(hp))] = v;
            }
        }
    };

    // ---- pipeline: one item of look-ahead.  Activation stage of DMA item q (q-th fetched chunk) = q & 1; the x_a item of a
    //      tile uses the stage of the tile's last shared chunk (free again after a barrier), while the other stage already
    //      receives the next tile's first chunk. -----------------------------------------------------------------------------
    issue_w(0, 0);
    issue_act(0);

    int n = 0;    // global item counter (weight stage = n & 1)
    int qd = 0;   // DMA'd chunks consumed so far
    for (int t = t_lo; t < t_hi; ++t) {
        int oy, ox;
        origin(t, &oy, &ox);
        // accumulators start at the bias (re-read per tile: 256 B from L1/L2, cheaper than 32 live registers)
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>((w < 2 ? p.bias_a : p.bias_b) + 16 * (w & 1) + 4 * sl);
#pragma unroll
            for (int row = 0; row < RPW; ++row)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) acc[row][w][ph] = bv;
        }
        // one pipeline item; BOTH = shared input chunk (feeds conv_a and conv_b), !BOTH = conv_b's x_a chunk
        // wait_dma: this wave's DMAs for the item are waited for here.  Not after an emit: gfx9's vmcnt counts stores too, so
        // a vmcnt(0) behind the emit's stores waits for their round trip to memory.  The items that follow an emit (the x_a
        // item; a tile's first item after the previous tile's x_b emit) have their DMAs waited for BEFORE that emit instead.
        auto run_item = [&](int j, auto both_tag, bool wait_dma) {
            constexpr bool BOTH = decltype(both_tag)::value;
            FW_STAMP(BOTH ? 1 : 2);   // previous phase ends (compute of the previous item / emit)
            if (wait_dma && !(FW_PAIR_DBG & 32)) FW_WAIT_VMCNT(0);
            FW_STAMP(5);              // waiting for this wave's own DMAs
            if (!(FW_PAIR_DBG & 16)) __syncthreads();
            FW_STAMP(0);              // barrier
            const bool more = n + 1 < nitems;
            const int jn = (j + 1 == ipt) ? 0 : j + 1;  // next item's position in its tile
            // next DMA'd chunk: item j+1 if it is a shared chunk; during the LAST shared chunk the next tile's first
            // chunk (the x_a item needs no DMA); during the x_a item nothing (both stages are occupied)
            const bool fetch = !(FW_PAIR_DBG & 1) && BOTH && more && (j + 1 < na || t + 1 < t_hi);
            const int fetch_stage = (qd + 1) & 1;
#ifndef FW_DMA_SLOT_W
#define FW_DMA_SLOT_W 2
#define FW_DMA_SLOT_A 6
#endif
#ifndef FW_DMA_SLOT_W2   // the younger half of the workgroup (waves NWAVES/2..): same slots unless told otherwise
#define FW_DMA_SLOT_W2 FW_DMA_SLOT_W
#define FW_DMA_SLOT_A2 FW_DMA_SLOT_A
#endif
            const bool young = wave >= NWAVES / 2;
            auto dma_slot = [&](int d) {
                if (d == (young ? FW_DMA_SLOT_W2 : FW_DMA_SLOT_W)) {
                    if (more && !(FW_PAIR_DBG & 2)) issue_w(jn, (n + 1) & 1);
                } else if (d == (young ? FW_DMA_SLOT_A2 : FW_DMA_SLOT_A)) {
                    if (fetch) issue_act(fetch_stage);
                }
            };
            // stage of this item: shared chunk -> qd & 1; x_a item -> the stage of the last shared chunk, (qd - 1) & 1
            const uint4* a = lds + ((BOTH ? qd : qd - 1) & 1) * ACT_REGION;
            const uint4* wl = lds + SM::W_BASE + (n & 1) * SM::W_REGION + lane;
            auto& slot_fn = dma_slot;
            conv_item<T, NW, BOTH ? 0 : 2>(acc, a, wl, rd_off, widx, slot_fn, [](const uint4 (&)[RPW][2]) {}, [&](int step) {
#ifdef FW_STEP_STAMPS  // BOTH items: slot 1 = steps 0-2 (the DMA slots), 6 = steps 3-5, 4 = steps 6-8
                if (BOTH && step == 2) FW_STAMP(1);
                if (BOTH && step == 5) FW_STAMP(6);
                if (BOTH && step == 8) FW_STAMP(4);
#endif
            });
            if (BOTH) ++qd;
            ++n;
        };
        FW_STAMP(4);  // tile setup (bias -> accumulators)
        bool wait_dma = t == t_lo;  // (a variable, and no unrolling: hipcc otherwise peels a second copy of the item body)
#pragma clang loop unroll(disable)
        for (int j = 0; j < na; ++j) {
            run_item(j, std::true_type{}, wait_dma);
            wait_dma = true;
        }
        FW_STAMP(1);
        // conv_a done for this wave.  In flight: the x_a item's weights and the next tile's first chunk, both issued an item
        // ago - waited for ahead of the stores (vmcnt counts stores too).
        if (!(FW_PAIR_DBG & 32)) FW_WAIT_VMCNT(0);
        FW_STAMP(5);
        // the whole 16x32 compute region inside the image?  (uniform; false only for border tiles)
        const bool interior = oy >= 0 && ox >= 0 && oy + TILE_H <= p.H && ox + TILE_W <= p.W;
        uint4 pk[RPW][2];
        if (!(FW_PAIR_DBG & 4)) {
            convert(0, pk);
            if (interior)
                store_out(pk, oy, ox, reinterpret_cast<T*>(p.out_a), std::true_type{});
            else
                store_out(pk, oy, ox, reinterpret_cast<T*>(p.out_a), std::false_type{});
        }
        FW_STAMP(6);  // emit: convert + stores
        // every wave must be finished with the last chunk's stage before it becomes the x_a tile
        __syncthreads();
        FW_STAMP(0);
        if (!(FW_PAIR_DBG & 4)) {
            if (interior)
                write_xa(pk, lds + ((qd - 1) & 1) * ACT_REGION, oy, ox, std::true_type{});
            else
                write_xa(pk, lds + ((qd - 1) & 1) * ACT_REGION, oy, ox, std::false_type{});
        }
        FW_STAMP(3);  // x_a tile into LDS
        run_item(na, std::false_type{}, false);
        FW_STAMP(2);  // x_a item compute
        // conv_b done for this wave.  In flight: the x_a stores (an item old) and the next tile's first weights.  No barrier:
        // the x_a tile's stage is refilled only behind the next item's barrier.
        if (!(FW_PAIR_DBG & 32)) FW_WAIT_VMCNT(0);
        FW_STAMP(5);
        if (!(FW_PAIR_DBG & 4)) {
            convert(2, pk);
            if (interior)
                store_out(pk, oy, ox, reinterpret_cast<T*>(p.out_b), std::true_type{});
            else
                store_out(pk, oy, ox, reinterpret_cast<T*>(p.out_b), std::false_type{});
        }
        FW_STAMP(6);
    }
    FW_STAMP_FLUSH(p.stamps);
}

bool pair_slide_enabled(int H, int W, int num_cus);
void launch_conv3x3_pair_slide(DType dt, const ConvPairParams& p, int num_cus, hipStream_t stream);
bool pair_slide32_enabled();   // conv3x3_pair_slide32.hip: the window kernel with all 32 columns of a tile valid
void launch_conv3x3_pair_slide32(DType dt, const ConvPairParams& p, int num_cus, hipStream_t stream);

static int pair_num_cus() {
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

void launch_conv3x3_pair(DType dt, const ConvPairParams& p_in, hipStream_t stream) {
    ConvPairParams p = p_in;
    if (p.H <= 0 || p.W <= 0 || p.na < 1 || p.na > 8) throw Error(1, "conv3x3_pair: bad problem");
    if (p.in_cstride < 32 || (p.in_cstride & 7) || p.in_pstride < 32 || (p.in_pstride & 7) || (p.out_cstride & 7) ||
        p.out_cstride < 32)
        throw Error(1, "conv3x3_pair: bad strides");
    if (!p.in || !p.wpk_a || !p.wpk_b || !p.bias_a || !p.bias_b || !p.out_a || !p.out_b) throw Error(1, "conv3x3_pair: NULL");
    p.zeros = conv_zero_page();
#ifdef FW_PAIR_STAMP
    p.stamps = stamp_buffer(0);
#endif
    if (pair_slide_enabled(p.H, p.W, pair_num_cus())) {  // the sliding-window form (conv3x3_pair_slide.hip) for all but small frames
        if (pair_slide32_enabled())
            launch_conv3x3_pair_slide32(dt, p, pair_num_cus(), stream);
        else
            launch_conv3x3_pair_slide(dt, p, pair_num_cus(), stream);
        return;
    }
    const int tiles = ((p.W + PAIR_TW - 1) / PAIR_TW) * ((p.H + PAIR_TH - 1) / PAIR_TH);
    dim3 grid(tiles < pair_num_cus() ? tiles : pair_num_cus()), block(64 * NWAVES);
    if (dt == DT_BF16)
        hipLaunchKernelGGL((conv3x3_pair_kernel<__bf16>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((conv3x3_pair_kernel<_Float16>), grid, block, 0, stream, p);
    FW_HIP_CHECK(hipGetLastError());
}

}  // namespace fw
