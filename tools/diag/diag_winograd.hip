// Diagnostic only (tools/winograd_kloop.py): the inner loop of the Winograd F(2x2, 3x3) trunk kernel DESIGN.md section 8 sketches,
// on static random f16 data in LDS - no DMA, no epilogue - to MEASURE what the paper design's K loop sustains before anything is
// built on it.  One "chunk" = 32 input channels of an 8 x 32-pixel output tile (64 Winograd tiles = 4 pixel tiles of 16) into 64
// output channels; wave i owns the four frequencies (i, l) of transform row i:
//   per pixel tile: 8 ds_read_b128 of the two raw halo rows the row transform needs (columns de-interleaved by parity so that 16
//   lanes on 16 consecutive tiles read 16 consecutive pieces), two v_pk_add_f16 stages -> 4 B fragments, 16 v_mfma_f32_16x16x32_f16
//   against the wave's 16 weight fragments (held in registers for the chunk).
// NW = 4: one wave per SIMD, 4 pixel tiles per wave, 256 accumulator registers; NW = 8: two waves per SIMD, 2 pixel tiles each.
// The direct kernel spends ~3500 cycles on the same 256 pixels x 32 channels x 64 outputs.
#include "fw_internal.h"
#include "conv_common.h"
#include "../../include/framewright_hip.h"

namespace {
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int RAW_PIECES = 10 * 4 * 2 * 17;   // [row 10][slot 4][parity 2][17] 16-byte pieces
constexpr int W_PIECES = 4 * 16 * 64;         // [frequency row 4][l 4][cout tile 4][lane 64]

template <int NW, bool SYNC>
__global__ __launch_bounds__(64 * NW, NW / 4) void winograd_kloop_kernel(const uint4* __restrict__ seed, int iters, float* sink,
                                                                         unsigned long long* clocks) {
    __shared__ __attribute__((aligned(16))) uint4 raw[RAW_PIECES];
    __shared__ __attribute__((aligned(16))) uint4 wl[W_PIECES];
    for (int i = threadIdx.x; i < RAW_PIECES; i += 64 * NW) raw[i] = seed[i & 1023];
    for (int i = threadIdx.x; i < W_PIECES; i += 64 * NW) wl[i] = seed[(i * 7 + 3) & 1023];
    __syncthreads();
    constexpr int PT = 16 / NW;               // pixel tiles per wave
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, q = lane & 15, sl = lane >> 4;
    const int fi = wave & 3, p0 = (wave >> 2) * PT;
    // rows of the 4x4 patch the row transform of frequency row fi combines: (0,2) (1,2) (2,1) (1,3); signs + - ... see below
    const int j1 = fi == 0 ? 0 : (fi == 2 ? 2 : 1), j2 = fi == 0 ? 2 : (fi == 1 ? 2 : (fi == 2 ? 1 : 3));
    const bool add = fi == 1;                 // r = d[j1] + d[j2] for row 1, d[j1] - d[j2] otherwise
    f4 acc[4][PT][4];
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[l][p][ct] = f4{0, 0, 0, 0};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        uint4 U[4][4];
#pragma unroll
        for (int l = 0; l < 4; ++l)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) U[l][ct] = wl[((fi * 4 + l) * 4 + ct) * 64 + lane];
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int rbase = 2 * (p0 + p);
            h8 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {     // column 2 q + k of rows rbase + j1 / j2: parity k & 1, index q + (k >> 1)
                const uint4 a = raw[(((rbase + j1) * 4 + sl) * 2 + (k & 1)) * 17 + q + (k >> 1)];
                const uint4 b = raw[(((rbase + j2) * 4 + sl) * 2 + (k & 1)) * 17 + q + (k >> 1)];
                const h8 ha = __builtin_bit_cast(h8, a), hb = __builtin_bit_cast(h8, b);
                r[k] = add ? ha + hb : ha - hb;
            }
            const h8 V[4] = {r[0] - r[2], r[1] + r[2], r[2] - r[1], r[1] - r[3]};
#pragma unroll
            for (int l = 0; l < 4; ++l)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[l][p][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, U[l][ct]), V[l], acc[l][p][ct], 0, 0, 0);
        }
        if (SYNC) __syncthreads();            // the raw tile of the next chunk would be published here
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float out = 0.f;
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) out += acc[l][p][ct][0];
    if (out == 123.456f) sink[0] = out;
    if (threadIdx.x == 0) {
        atomicAdd(clocks, c1 - c0);
        atomicAdd(clocks + 1, r1 - r0);
    }
}

// The same K loop with its data movement: four waves per CU; per chunk every wave re-fetches its own 16-KiB weight slice by LDS-DMA
// from a 384-KiB (L2-resident) buffer as soon as the slice sits in its registers, and the workgroup streams the next 21.25-KiB raw
// tile from a large (HBM) buffer into the other raw buffer; one barrier per chunk publishes the raw tile.  No epilogue.
constexpr int RAW_KIB = 22;                      // raw tile padded to 22 KiB in the source buffer
// WREG: the weight slices go straight from L2 into registers with plain 16-byte loads (one chunk ahead) instead of through LDS-DMA.
template <bool WREG>
__global__ __launch_bounds__(256, 1) void winograd_stream_kernel(const char* __restrict__ wsrc, const char* __restrict__ rsrc, long rtiles, int iters,
                                                                 int stagger /* workgroup b starts at weight chunk b % 6 */, float* sink,
                                                                 unsigned long long* clocks) {
    __shared__ __attribute__((aligned(16))) uint4 raw[2][RAW_KIB * 64];
    __shared__ __attribute__((aligned(16))) uint4 wl[W_PIECES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, q = lane & 15, sl = lane >> 4;
    const unsigned raw_lds = (unsigned)(size_t)(fw::lds_ptr_t)raw, w_lds = (unsigned)(size_t)(fw::lds_ptr_t)wl;
    const int fi = wave;
    const int j1 = fi == 0 ? 0 : (fi == 2 ? 2 : 1), j2 = fi == 0 ? 2 : (fi == 1 ? 2 : (fi == 2 ? 1 : 3));
    const bool add = fi == 1;
    const int cph = stagger ? (int)(blockIdx.x % 6) : 0;
    auto fetch_w = [&](int chunk) {            // this wave's slice of chunk `chunk % 6`: 16 KiB in four batches of four pieces
        const char* src = wsrc + ((size_t)((chunk + cph) % 6) * 4 + fi) * 16384;
#pragma unroll
        for (int k = 0; k < 4; ++k) fw::glds16_batch_w<4>(src + 4096 * (k + 1), lane * 16, w_lds + fi * 16384 + 4096 * (k + 1));
    };
    auto fetch_raw = [&](long tile, int buf) { // pieces wave, wave + 4, ... of the 22 KiB
        const char* src = rsrc + (size_t)(tile % rtiles) * (RAW_KIB * 1024);
        for (int k = wave; k < RAW_KIB; k += 4) fw::glds16(src + k * 1024, lane * 16, raw_lds + buf * (RAW_KIB * 1024) + k * 1024);
    };
    f4 acc[4][4][4];
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[l][p][ct] = f4{0, 0, 0, 0};
    const long t0 = (long)blockIdx.x * iters;
    uint4 Un[4][4];
    auto load_w = [&](int chunk) {
        const uint4* src = reinterpret_cast<const uint4*>(wsrc + ((size_t)((chunk + cph) % 6) * 4 + fi) * 16384) + lane;
#pragma unroll
        for (int l = 0; l < 4; ++l)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) Un[l][ct] = src[(l * 4 + ct) * 64];
    };
    if (WREG) load_w(0); else fetch_w(0);
    fetch_raw(t0, 0);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        FW_WAIT_VMCNT(0);                       // this wave's slice and its pieces of raw tile `it` have landed
        __syncthreads();                        // ... and everybody else's pieces
        uint4 U[4][4];
        if (WREG) {
#pragma unroll
            for (int l = 0; l < 4; ++l)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) U[l][ct] = Un[l][ct];
            if (it + 1 < iters) {
                load_w(it + 1);
                fetch_raw(t0 + it + 1, (it + 1) & 1);
            }
        } else {
#pragma unroll
            for (int l = 0; l < 4; ++l)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) U[l][ct] = wl[((fi * 4 + l) * 4 + ct) * 64 + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slice is in registers: it may be overwritten
            if (it + 1 < iters) {
                fetch_w(it + 1);
                fetch_raw(t0 + it + 1, (it + 1) & 1);
            }
        }
        const uint4* rw = raw[it & 1];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int rbase = 2 * p;
            h8 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 a = rw[(((rbase + j1) * 4 + sl) * 2 + (k & 1)) * 17 + q + (k >> 1)];
                const uint4 b = rw[(((rbase + j2) * 4 + sl) * 2 + (k & 1)) * 17 + q + (k >> 1)];
                const h8 ha = __builtin_bit_cast(h8, a), hb = __builtin_bit_cast(h8, b);
                r[k] = add ? ha + hb : ha - hb;
            }
            const h8 V[4] = {r[0] - r[2], r[1] + r[2], r[2] - r[1], r[1] - r[3]};
#pragma unroll
            for (int l = 0; l < 4; ++l)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[l][p][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, U[l][ct]), V[l], acc[l][p][ct], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float out = 0.f;
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) out += acc[l][p][ct][0];
    if (out == 123.456f) sink[0] = out;
    if (threadIdx.x == 0) {
        atomicAdd(clocks, c1 - c0);
        atomicAdd(clocks + 1, r1 - r0);
    }
}
}  // namespace

// the streaming variant: `iters` chunks per workgroup, raw tiles from a buffer of `raw_mib` MiB (0: one tile re-read: L2), ms + clocks
extern "C" int fw_debug_winograd_stream(int wreg, int stagger, int blocks, int iters, int raw_mib, float* ms_out, unsigned long long* clocks_out) {
    if (blocks < 1 || iters < 1 || raw_mib < 0 || !ms_out || !clocks_out) return FW_ERR_INVALID;
    const size_t wbytes = 6 * 65536, rbytes = raw_mib > 0 ? (size_t)raw_mib << 20 : (size_t)RAW_KIB * 1024;
    const long rtiles = (long)(rbytes / (RAW_KIB * 1024));
    char *w = nullptr, *r = nullptr;
    float* sink = nullptr;
    unsigned long long* clk = nullptr;
    if (hipMalloc((void**)&w, wbytes) != hipSuccess || hipMalloc((void**)&r, rbytes) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess ||
        hipMalloc((void**)&clk, 16) != hipSuccess)
        return FW_ERR_OOM;
    (void)hipMemset(w, 0x3c, wbytes);          // 0x3c3c: f16 1.0586
    (void)hipMemset(r, 0x38, rbytes);          // 0x3838: f16 0.527
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) {
            (void)hipMemset(clk, 0, 16);
            (void)hipEventRecord(e0, nullptr);
        }
        if (wreg) hipLaunchKernelGGL(winograd_stream_kernel<true>, dim3(blocks), dim3(256), 0, nullptr, (const char*)w, (const char*)r, rtiles, iters, stagger, sink, clk);
        else hipLaunchKernelGGL(winograd_stream_kernel<false>, dim3(blocks), dim3(256), 0, nullptr, (const char*)w, (const char*)r, rtiles, iters, stagger, sink, clk);
    }
    (void)hipEventRecord(e1, nullptr);
    const int rc = hipEventSynchronize(e1) == hipSuccess ? FW_OK : FW_ERR_HIP;
    (void)hipEventElapsedTime(ms_out, e0, e1);
    (void)hipMemcpy(clocks_out, clk, 16, hipMemcpyDeviceToHost);
    (void)hipFree(w);
    (void)hipFree(r);
    (void)hipFree(sink);
    (void)hipFree(clk);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

namespace {
}  // namespace

// `iters` chunks on `blocks` workgroups of `waves` (4 or 8) waves; ms of the launch and the summed s_memtime / s_memrealtime deltas
// of thread 0 of every workgroup
extern "C" int fw_debug_winograd_kloop(int waves, int sync, int blocks, int iters, float* ms_out, unsigned long long* clocks_out) {
    if ((waves != 4 && waves != 8) || blocks < 1 || iters < 1 || !ms_out || !clocks_out) return FW_ERR_INVALID;
    uint4* seed = nullptr;
    float* sink = nullptr;
    unsigned long long* clk = nullptr;
    if (hipMalloc((void**)&seed, 1024 * 16) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess || hipMalloc((void**)&clk, 16) != hipSuccess)
        return FW_ERR_OOM;
    uint16_t h[8192];
    unsigned s = 12345u;
    for (int i = 0; i < 8192; ++i) {          // random f16 in roughly [-2, 2)
        s = s * 1664525u + 1013904223u;
        h[i] = (uint16_t)(((s >> 16) & 0x8000u) | ((13u + ((s >> 8) & 3u)) << 10) | ((s >> 20) & 0x3ffu));
    }
    (void)hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) {
            (void)hipMemset(clk, 0, 16);
            (void)hipEventRecord(e0, nullptr);
        }
        if (waves == 4 && sync) hipLaunchKernelGGL((winograd_kloop_kernel<4, true>), dim3(blocks), dim3(256), 0, nullptr, seed, iters, sink, clk);
        else if (waves == 4) hipLaunchKernelGGL((winograd_kloop_kernel<4, false>), dim3(blocks), dim3(256), 0, nullptr, seed, iters, sink, clk);
        else if (sync) hipLaunchKernelGGL((winograd_kloop_kernel<8, true>), dim3(blocks), dim3(512), 0, nullptr, seed, iters, sink, clk);
        else hipLaunchKernelGGL((winograd_kloop_kernel<8, false>), dim3(blocks), dim3(512), 0, nullptr, seed, iters, sink, clk);
    }
    (void)hipEventRecord(e1, nullptr);
    const int rc = hipEventSynchronize(e1) == hipSuccess ? FW_OK : FW_ERR_HIP;
    (void)hipEventElapsedTime(ms_out, e0, e1);
    (void)hipMemcpy(clocks_out, clk, 16, hipMemcpyDeviceToHost);
    (void)hipFree(seed);
    (void)hipFree(sink);
    (void)hipFree(clk);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
