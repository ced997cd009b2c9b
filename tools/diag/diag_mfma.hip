// Diagnostic only (tools/mfma_peak.py): what the matrix cores of THIS chip sustain on random bf16 operands when nothing else
// is in the way — a bare v_mfma_f32_16x16x32_bf16 / 32x32x16 loop, operands in registers, 1 or 2 waves per SIMD on every CU.
// DESIGN.md §6 quotes the result next to the kernels' achieved rate: the chip lowers its clock under such a loop, so the
// spec-sheet 2.5 PFLOP/s is not what a kernel can reach on real data.
#include "fw_internal.h"
#include "../../include/framewright_hip.h"

namespace {
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

// ORDER (16x16x32 only): which operand stays while the other cycles through its four registers - 0: B stays for four MFMAs (A changes with
// every MFMA), 1: A stays, 2: both change with every MFMA, 3: one A and one B register for all of them.  (Does the order in which a
// kernel walks its fragments change what the chip can sustain at the power cap?)
template <int SHAPE, int ORDER = 0>
__global__ __launch_bounds__(512, 2) void mfma_peak_kernel(const uint4* __restrict__ seed, int iters, float* sink,
                                                           unsigned long long* clocks) {
    const int lane = threadIdx.x & 63;
    b8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(b8, seed[(lane + 64 * i) & 1023]);
        b[i] = __builtin_bit_cast(b8, seed[(lane + 64 * (i + 4)) & 1023]);
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float out = 0.f;
    if constexpr (SHAPE == 16) {
        f4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ORDER == 0 ? (i & 3) : ORDER == 1 ? (i >> 2) : ORDER == 2 ? (i & 3) : 0],
                                                                 b[ORDER == 0 ? (i >> 2) : ORDER == 1 ? (i & 3) : ORDER == 2 ? ((i + (i >> 2)) & 3) : 0], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) out += acc[i][0];
    } else {
        f16v acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + r) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) out += acc[i][0];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (out == 123.456f) sink[0] = out;   // keep the accumulators alive
    if (threadIdx.x == 0) {
        atomicAdd(clocks, c1 - c0);
        atomicAdd(clocks + 1, r1 - r0);
    }
}
}  // namespace

// Runs `iters` x 16 (shape 16) or x 8 (shape 32) MFMAs per wave on `blocks` workgroups of 8 waves; returns milliseconds of the
// launch (HIP events), the summed s_memtime / s_memrealtime deltas of wave 0 of every workgroup in clocks[0..1].
extern "C" int fw_debug_mfma_peak(int shape, int order, int zeros, int blocks, int iters, float* ms_out, unsigned long long* clocks_out) {
    if ((shape != 16 && shape != 32) || order < 0 || order > 3 || blocks < 1 || iters < 1 || !ms_out || !clocks_out) return FW_ERR_INVALID;
    uint4* seed = nullptr;
    float* sink = nullptr;
    unsigned long long* clk = nullptr;
    if (hipMalloc((void**)&seed, 1024 * 16) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess ||
        hipMalloc((void**)&clk, 16) != hipSuccess)
        return FW_ERR_OOM;
    uint16_t h[8192];
    unsigned s = 12345u;
    for (int i = 0; i < 8192; ++i) {   // random bf16 in roughly [-2, 2): sign, exponent 125..128, random mantissa
        s = s * 1664525u + 1013904223u;
        h[i] = (uint16_t)(((s >> 16) & 0x8000u) | ((125u + ((s >> 8) & 3u)) << 7) | ((s >> 20) & 0x7fu));
    }
    if (zeros) memset(h, 0, sizeof(h));   // all-zero operands: nothing toggles in the multipliers
    (void)hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    (void)hipMemset(clk, 0, 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {   // first launch warms up, second is timed
        if (rep == 1) {
            (void)hipMemset(clk, 0, 16);
            (void)hipEventRecord(e0, nullptr);
        }
        if (shape == 16 && order == 1)
            hipLaunchKernelGGL((mfma_peak_kernel<16, 1>), dim3(blocks), dim3(512), 0, nullptr, seed, iters, sink, clk);
        else if (shape == 16 && order == 2)
            hipLaunchKernelGGL((mfma_peak_kernel<16, 2>), dim3(blocks), dim3(512), 0, nullptr, seed, iters, sink, clk);
        else if (shape == 16 && order == 3)
            hipLaunchKernelGGL((mfma_peak_kernel<16, 3>), dim3(blocks), dim3(512), 0, nullptr, seed, iters, sink, clk);
        else if (shape == 16)
            hipLaunchKernelGGL(mfma_peak_kernel<16>, dim3(blocks), dim3(512), 0, nullptr, seed, iters, sink, clk);
        else
            hipLaunchKernelGGL(mfma_peak_kernel<32>, dim3(blocks), dim3(512), 0, nullptr, seed, iters, sink, clk);
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
