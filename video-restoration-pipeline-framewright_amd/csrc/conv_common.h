// Device-side pieces shared by the conv3x3 MFMA kernels (conv3x3_mfma.hip, conv3x3_pair.hip).  gfx950 only.
#pragma once
#include "fw_internal.h"

namespace fw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Op;
template <>
struct Op<__bf16> {
    // D(16 cout x 16 px) += A(16 cout x 32 cin) * B(32 cin x 16 px); lane l: A row / B,D column l & 15, k = 8*(l >> 4) + j
    // (16 bytes), D rows 4*(l >> 4) + j
    static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
        bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
        return __builtin_bit_cast(uint2, v);
    }
    static __device__ __forceinline__ f32x4 unpack4(uint2 v) {
        return f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                     __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u)};
    }
};
template <>
struct Op<_Float16> {
    static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
        f16x4 v = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
        return __builtin_bit_cast(uint2, v);
    }
    static __device__ __forceinline__ f32x4 unpack4(uint2 v) {
        const f16x4 q = __builtin_bit_cast(f16x4, v);
        return f32x4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
    }
};

// LeakyReLU(0.2) = max(v, 0.2 v).  v_max_f32 through inline asm: the builtin max puts a canonicalising v_max v, v, v in
// front of every element (IEEE mode), 4 of the 10 VALU instructions per four values in the emit phases.
__device__ __forceinline__ f32x4 lrelu4(f32x4 v) {
    const f32x4 t = v * 0.2f;
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) asm("v_max_f32 %0, %1, %2" : "=v"(o[j]) : "v"(v[j]), "v"(t[j]));
    return o;
}

// Output stores.  FW_NT_STORES: as non-temporal (streaming) stores - the outputs are never read again by the kernel that
// writes them, and whatever sits dirty in the XCDs' L2s when a kernel ends is written back before the next one starts.
#ifndef FW_NT_STORES
#define FW_NT_STORES 0
#endif
typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(void* dst, uint4 v) {
#if FW_NT_STORES
    __builtin_nontemporal_store(nt_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_u32x4*>(dst));
#else
    *reinterpret_cast<uint4*>(dst) = v;
#endif
}

constexpr int TILE_H = 16;
constexpr int TILE_W = 32;
constexpr int HALO_H = TILE_H + 2;                        // 18
constexpr int HALO_W = TILE_W + 2;                        // 34
constexpr int ROW_PIECES = HALO_W * 4;                    // 136 16-byte pieces per halo row
constexpr int ACT_PIECES = HALO_H * ROW_PIECES;           // 2448 pieces per 32-channel chunk
constexpr int ACT_INSTR = 40;                            // wave-instructions of 1 KiB per chunk (39 used + 1 pad)
constexpr int ACT_REGION = ACT_INSTR * 64;                // 2560 pieces = 40 KiB per stage
#ifndef FW_NWAVES
#define FW_NWAVES 8
#endif
constexpr int NWAVES = FW_NWAVES;                         // 8: two waves per SIMD hide each other's stalls; 4: one per SIMD,
                                                          // four rows per wave, 40 % fewer LDS read bytes per MFMA
static_assert(NWAVES == 4 || NWAVES == 8, "waves per workgroup");
constexpr int WAVES_PER_SIMD = NWAVES / 4;
constexpr int RPW = TILE_H / NWAVES;                      // output rows per wave (2)
constexpr int ACT_ITERS = ACT_INSTR / NWAVES;             // 5 per wave, every wave issues all of them
constexpr int W_FRAGS = 18;                               // 9 taps x 2 k-steps of 16, per cout tile

// LDS image of one activation chunk: [halo row][halo px][4 slots of 16 B]; slot s (= 8 channels) of pixel p is stored
// at slot s ^ halo_swz(p).  A B-fragment read (v_mfma_f32_16x16x32: lane l -> pixel p0 + (l & 15), slot l >> 4) puts the
// four 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) on 16 distinct 16-byte bank positions
// (p & 3) * 4 + (s ^ halo_swz(p)) for every p0 - the swizzle 2*((p >> 2) & 1) is the one that does (the others with
// period <= 8 pixels were enumerated and conflict).
// The image is filled by LDS-DMA (global_load_lds_dwordx4): the LDS destination of a wave-instruction is
// lane-linear, so the swizzle is applied on the per-lane SOURCE address (cdna_hip_programming.md rule 21).
// Lanes whose halo position is outside the image read a 16-byte zero page instead, so every wave issues the same
// number of DMA instructions per stage (the counted s_waitcnt vmcnt below relies on it) and zero padding costs
// nothing.
// Tile walk.  A workgroup owns a contiguous range of the linear tile index; FW_TILE_ORDER says how that index maps to the
// grid: 0 = row-major (a workgroup walks a horizontal run), 1 = column-major (a vertical run: a tile shares 4 of its 18
// halo rows with the tile the same workgroup did just before, and its left / right neighbours are being done at the same
// time by workgroups of the same XCD).
#ifndef FW_TILE_ORDER
#define FW_TILE_ORDER 1
#endif
__device__ __forceinline__ void tile_pos(int t, int tiles_x, int tiles_y, int* ty, int* tx) {
    if (FW_TILE_ORDER == 0) {
        *ty = t / tiles_x;
        *tx = t - *ty * tiles_x;
    } else {
        *tx = t / tiles_y;
        *ty = t - *tx * tiles_y;
    }
}

__host__ __device__ constexpr int halo_swz(int px) { return ((px >> 2) & 1) << 1; }

template <int CT>
struct Smem {
    static constexpr int NA = CT == 1 ? 3 : 2;                  // activation stages in flight
    static constexpr int W_ITERS = (W_FRAGS * CT + NWAVES - 1) / NWAVES;  // weight DMAs per wave per stage (<= 3 or 5)
    static constexpr int W_REGION = W_FRAGS * CT * 64;          // pieces per weight stage
    static constexpr int W_BASE = NA * ACT_REGION;
    static constexpr int TOTAL = NA * ACT_REGION + 2 * W_REGION;  // CT=1: 156 KiB; CT=2: 152 KiB
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// One global_load_lds_dwordx4: every active lane copies 16 bytes from sbase + voff (wave-uniform 64-bit base in an SGPR
// pair, 32-bit unsigned per-lane byte offset) to LDS[lds_dst + 16 * lane]; lds_dst is wave-uniform.  The SGPR-base form
// keeps the issuing wave's VALU out of it: the per-lane offsets are loop constants and the base moves with scalar adds.
// Inline asm so that hipcc does not count these loads: with the builtin it drains them (s_waitcnt vmcnt(0)) before the
// first ds_read of the chunk being computed, which serialises the pipeline (cdna_hip_programming.md §5 "Three .s-level
// traps" (b)).  The matching waits are the explicit counted vmcnt at the top of the chunk loop.  M0 is written inside the
// statement; nothing else in these kernels uses M0.  Inactive lanes (EXEC) neither load nor write LDS.
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    // wave-uniform by construction; readfirstlane makes it provably so where the compiler has lost track (SGPR operands)
    const unsigned long long b = (unsigned long long)sbase;
    // (the builtin returns int: without the casts the low half would be sign-extended over the high half)
    const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b);
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    // s_nop 4: an SGPR written by v_readfirstlane must not be read as the base of a global_* for 5 wait states, and
    // nothing inside an asm statement is padded by hipcc
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sb), "s"(lds_dst)
        : "memory");
}

// Batches: ONE M0 write for up to five pieces of a wave.  The instruction's 13-bit signed immediate offset is added to the
// global address AND to the LDS address (checked on gfx950: test_conv3x3_gpu passes with pieces told apart only by it), so
// pieces whose LDS images are 1 KiB apart share an M0 that points at the fifth piece: piece i sits at offset (i - 4) KiB.
// Rewriting M0 between LDS-DMAs is what made each of them cost the issuing wave 200-400 cycles (phase stamps, DESIGN.md
// §6); a batch costs about one.
// Same per-lane offset for every piece (weights: the source is as linear as the LDS image).
template <int N>
__device__ __forceinline__ void glds16_batch_w(const void* sbase, unsigned voff, unsigned lds_piece4) {
    static_assert(N >= 3 && N <= 5, "batch sizes in use");
    const unsigned long long b = (unsigned long long)sbase;
    const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b);
    lds_piece4 = __builtin_amdgcn_readfirstlane(lds_piece4);
    if constexpr (N == 5)
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:-4096\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:-3072\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:-2048\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:-1024\n\t"
            "global_load_lds_dwordx4 %0, %1"
            :
            : "v"(voff), "s"(sb), "s"(lds_piece4)
            : "memory");
    else if constexpr (N == 4)
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:-4096\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:-3072\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:-2048\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:-1024"
            :
            : "v"(voff), "s"(sb), "s"(lds_piece4)
            : "memory");
    else
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:-4096\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:-3072\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:-2048"
            :
            : "v"(voff), "s"(sb), "s"(lds_piece4)
            : "memory");
}
// Five pieces with their own per-lane offsets (activations); voff[i] already contains -(i - 4) KiB.
__device__ __forceinline__ void glds16_batch_a(const void* sbase, const unsigned* voff, unsigned lds_piece4) {
    const unsigned long long b = (unsigned long long)sbase;
    const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b);
    lds_piece4 = __builtin_amdgcn_readfirstlane(lds_piece4);
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %5 offset:-4096\n\t"
        "global_load_lds_dwordx4 %1, %5 offset:-3072\n\t"
        "global_load_lds_dwordx4 %2, %5 offset:-2048\n\t"
        "global_load_lds_dwordx4 %3, %5 offset:-1024\n\t"
        "global_load_lds_dwordx4 %4, %5"
        :
        : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "s"(sb), "s"(lds_piece4)
        : "memory");
}

// Six pieces with their own per-lane offsets: the fifth piece is M0's, the sixth sits one KiB above it; voff[i] already contains
// -(i - 4) KiB (conv3x3_pair_slide32.hip: 43 KiB per stage over eight waves = five waves of 5 and three of 6).
__device__ __forceinline__ void glds16_batch_a6(const void* sbase, const unsigned* voff, unsigned lds_piece4) {
    const unsigned long long b = (unsigned long long)sbase;
    const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b);
    lds_piece4 = __builtin_amdgcn_readfirstlane(lds_piece4);
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 m0, %7\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %6 offset:-4096\n\t"
        "global_load_lds_dwordx4 %1, %6 offset:-3072\n\t"
        "global_load_lds_dwordx4 %2, %6 offset:-2048\n\t"
        "global_load_lds_dwordx4 %3, %6 offset:-1024\n\t"
        "global_load_lds_dwordx4 %4, %6\n\t"
        "global_load_lds_dwordx4 %5, %6 offset:1024"
        :
        : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]), "s"(sb), "s"(lds_piece4)
        : "memory");
}

// Four pieces with their own per-lane offsets (the GEMM kernel's activation rows); voff[i] already contains -(i - 3) KiB and
// lds_piece3 is the LDS address of the FOURTH piece.
__device__ __forceinline__ void glds16_batch_a4(const void* sbase, const unsigned* voff, unsigned lds_piece3) {
    const unsigned long long b = (unsigned long long)sbase;
    const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b);
    lds_piece3 = __builtin_amdgcn_readfirstlane(lds_piece3);
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %4 offset:-3072\n\t"
        "global_load_lds_dwordx4 %1, %4 offset:-2048\n\t"
        "global_load_lds_dwordx4 %2, %4 offset:-1024\n\t"
        "global_load_lds_dwordx4 %3, %4"
        :
        : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(sb), "s"(lds_piece3)
        : "memory");
}

// The same transfer with a per-lane 64-bit source address (border tiles: pieces outside the image read the zero page, so
// every wave still issues exactly one DMA per piece - the counted vmcnt waits rely on that).
__device__ __forceinline__ void glds16_v(const void* gsrc, unsigned lds_dst) {
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// The 18 weight fragments of one 32-output-channel half, split over the NWAVES/2 waves that own the half: 8 waves -> 5, 5,
// 5, 3 fragments; 4 waves -> 9 each, as batches of 5 + 4.  `src`/`dst` address fragment 0 of the half (global / LDS byte
// address), k = the wave's index within the half.
__device__ __forceinline__ void issue_w_half(const char* src, unsigned dst, int k, unsigned lane16) {
    if constexpr (NWAVES == 8) {
        const int f4 = 5 * k + 4;  // the batch's fifth fragment
        if (k < 3)
            glds16_batch_w<5>(src + f4 * 1024, lane16, dst + f4 * 1024);
        else
            glds16_batch_w<3>(src + f4 * 1024, lane16, dst + f4 * 1024);
    } else {
        const int f4 = 9 * k + 4;
        glds16_batch_w<5>(src + f4 * 1024, lane16, dst + f4 * 1024);
        glds16_batch_w<4>(src + (f4 + 5) * 1024, lane16, dst + (f4 + 5) * 1024);
    }
}

// Phase stamps (diagnostic build -DFW_PAIR_STAMP only; in the product build no stamp executes): every wave accumulates
// s_memtime deltas per phase into 8 slots; slot 7 = the wave's lifetime in 100 MHz s_memrealtime ticks, so the clock the
// chip held is sum(slots 0..6) / slot 7 * 100 MHz.  stamp_buffer(k): 8 waves x 8 slots of the pair kernel (k = 0) and the
// 64-channel residual conv (k = 1).
#ifdef FW_PAIR_STAMP
#define FW_STAMP_INIT()                                                        \
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};                  \
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();              \
    const unsigned long long stamp_rt0 = __builtin_amdgcn_s_memrealtime()
#define FW_STAMP(slot)                                                   \
    do {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                               \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();      \
        __builtin_amdgcn_sched_barrier(0);                               \
        stamp_acc[slot] += t_ - stamp_last;                              \
        stamp_last = t_;                                                 \
    } while (0)
#define FW_STAMP_FLUSH(buf)                                                                       \
    do {                                                                                          \
        stamp_acc[7] = __builtin_amdgcn_s_memrealtime() - stamp_rt0;                              \
        if ((threadIdx.x & 63) == 0 && (buf)) {                                                   \
            for (int k_ = 0; k_ < 8; ++k_) atomicAdd((buf) + (threadIdx.x >> 6) * 8 + k_, stamp_acc[k_]); \
            /* entries 64 + wave: the SLOWEST workgroup's lifetime, summed over launches via max-per-launch is not */ \
            /* possible with one atomic: keep the overall max and the sum of squares for a spread estimate */       \
            atomicMax((buf) + 64 + (threadIdx.x >> 6), stamp_acc[7]);                             \
            atomicAdd((buf) + 72 + (threadIdx.x >> 6), stamp_acc[7] * stamp_acc[7]);              \
        }                                                                                         \
    } while (0)
unsigned long long* stamp_buffer(int which);
#else
#define FW_STAMP_INIT() do { } while (0)
#define FW_STAMP(slot) do { } while (0)
#define FW_STAMP_FLUSH(buf) do { } while (0)
#endif

// Counted wait on the vector-memory counter as a BUILTIN (simm16: vmcnt | expcnt 7 << 4 | lgkmcnt 15 << 8), not inline asm:
// hipcc's waitcnt pass then knows that its own loads (bias, residual planes) have landed and does not wait for them again
// together with whatever was issued after them.  The LDS-DMAs themselves stay invisible to it (glds16).
#define FW_WAIT_VMCNT(n)                                   \
    do {                                                   \
        __builtin_amdgcn_s_waitcnt(0x0F70 | (n));          \
        asm volatile("" ::: "memory");                     \
    } while (0)

#ifdef FW_NO_SB
#define FW_SB()
#else
#define FW_SB() __builtin_amdgcn_sched_barrier(0)
#endif

// One pipeline item: 32 input channels x 9 taps of the wave's RPW rows x 32 pixels into NW - W_LO accumulator tiles of
// 16 output channels (x 2 pixel halves).  `a` = the activation stage, `wl` = the weight stage + lane, widx(tap, w) = the
// fragment index of weight tile w at that tap, rd_off[dx][ph] = the lane's piece offset in halo row RPW*wave.
// Schedule: 9 steps (dx outer, dy inner) x (NW - W_LO) weight tiles, 4 MFMAs (RPW rows x 2 pixel halves) per weight tile.
// Weight fragments run through a 3-deep register ring, read two tiles ahead.  The B fragments roll in place: rows 0/1 of
// the next dx are read into their registers as soon as dy = 1/2 has issued (their last use), rows 2/3 at the start of the
// next dx, one step before they are needed.  sched_barrier(0) pins that order; left alone hipcc sinks every ds_read to just
// before its first use.
// dma_slot(k), k = 0..35, is called after every 2*(NW - W_LO) MFMAs; on_centre(xc) once with the centre-tap B fragments
// xc[row][ph] (the wave's own pixels).
// conv_item_rp: the same with ROWP pieces per halo row of the stage (the 36-pixel stages of conv3x3_pair_slide32.hip: 145).
template <int ROWP, typename T, int NW, int W_LO, typename WIdx, typename Slot, typename Centre, typename Hook>
__device__ __forceinline__ void conv_item_rp(f32x4 (&acc)[RPW][NW][2], const uint4* a, const uint4* wl, const int (&rd_off)[3][2],
                                             WIdx widx, Slot dma_slot, Centre on_centre, Hook step_hook) {
    constexpr int NWU = NW - W_LO;
    constexpr int NK = 9 * NWU;  // weight fragments of the item, in use order
    uint4 xr[RPW + 2][2];
#ifndef FW_WRING
#define FW_WRING 3
#endif
    constexpr int RING = FW_WRING;   // weight-fragment register ring: fragments are read RING - 1 tiles ahead
    uint4 wf[RING];
    auto load_w = [&](int k) {
        const int t = k / NWU, w = k - t * NWU;
        const int dx = t / 3, dy = t - 3 * dx;
        wf[k % RING] = wl[widx(dy * 3 + dx, W_LO + w) * 64];
    };
#pragma unroll
    for (int row = 0; row < RPW + 2; ++row)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) xr[row][ph] = a[row * ROWP + rd_off[0][ph]];
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) load_w(k);
    FW_SB();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dx = t / 3, dy = t - 3 * dx;
        if (dx < 2 && dy >= 1) {
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) xr[dy - 1][ph] = a[(dy - 1) * ROWP + rd_off[dx + 1][ph]];
        }
        if (dx > 0 && dy == 0) {
#pragma unroll
            for (int row = 2; row < RPW + 2; ++row)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) xr[row][ph] = a[row * ROWP + rd_off[dx][ph]];
        }
#pragma unroll
        for (int w = 0; w < NWU; ++w) {
            const int k = t * NWU + w;
            if (k + RING - 1 < NK) load_w(k + RING - 1);
            FW_SB();
#pragma unroll
            for (int row = 0; row < RPW; ++row)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph)
                    acc[row][W_LO + w][ph] = Op<T>::mfma16(wf[k % RING], xr[row + dy][ph], acc[row][W_LO + w][ph]);
            FW_SB();
            // 36 slots per item: NWU = 4 -> one per weight tile; NWU = 2 -> two per weight tile
#pragma unroll
            for (int d = 0; d < 4 / NWU; ++d) dma_slot((t * NWU + w) * (4 / NWU) + d);
            FW_SB();
        }
        if (dx == 1 && dy == 1) {
            uint4 xc[RPW][2];
#pragma unroll
            for (int row = 0; row < RPW; ++row)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) xc[row][ph] = xr[row + 1][ph];
            on_centre(xc);
            FW_SB();
        }
        step_hook(t);  // diagnostics only (phase stamps)
    }
}

template <typename T, int NW, int W_LO, typename WIdx, typename Slot, typename Centre, typename Hook>
__device__ __forceinline__ void conv_item(f32x4 (&acc)[RPW][NW][2], const uint4* a, const uint4* wl, const int (&rd_off)[3][2],
                                          WIdx widx, Slot dma_slot, Centre on_centre, Hook step_hook) {
    conv_item_rp<ROW_PIECES, T, NW, W_LO>(acc, a, wl, rd_off, widx, dma_slot, on_centre, step_hook);
}

}  // namespace fw
