// The second half of a 128-channel NAFBlock in one pass over the pixels (reference: the NAFNet the TAP driver loads,
// tap_denoise.py:335-364; block layout as in oracle/nafnet_ref.py):
//
//     y   = inp + beta  * conv3(x * sca)            x: the gated tensor of the block's first half, typed [M][128]
//     out = y   + gamma * conv5(SimpleGate(conv4(LayerNorm2d(y))))
//
// As four kernels (conv3, LayerNorm, conv4 + gate, conv5) the level moved 36 c bytes per pixel and block, 30 c of them here; this
// kernel reads x (2 c) and the fp32 stream (4 c) and writes the stream (4 c).  nn_ops.hip's naf_tail64_kernel does the same at 64
// channels with every weight matrix resident in LDS; at 128 channels they are 128 KB, so they stream:
//
//   * a persistent 512-thread workgroup walks 256-pixel tiles; wave w owns pixels [32 w, 32 w + 32) of the tile - two MFMA
//     pixel tiles - for the whole chain: a pixel never leaves its lane quartet between the three GEMMs;
//   * the weights arrive as eight 17-KiB blocks per tile (LDS-DMA, two buffers, one barrier per block): conv3 in two blocks of
//     64 output channels, conv4 in four blocks of 32 x1 + the 32 x2 channels they are gated with, conv5 in two blocks; a block's
//     tail carries its bias and its beta / gamma factors.  The SCA factor is folded into a scaled copy of the two conv3 blocks
//     per forward (naf_tail128_scale_w3_kernel), LayerNorm2d's affine part into conv4 by the host;
//   * D fragments become the next GEMM's B fragments without leaving the registers: the K order of the packed conv4 / conv5
//     weights is the accumulator layout (lane (q, s) of v_mfma_f32_16x16x32 holds channels 16 ct + 4 s + i of pixel q): chunk
//     kc, element e  <->  channel 16 (2 kc + (e >> 2)) + 4 s + (e & 3);
//   * y stays in 64 fp32 registers per lane from conv3's epilogue to conv5's.
#include "fw_internal.h"
#include "conv_common.h"

namespace fw {

constexpr int T128_PX = 256;                       // pixels per workgroup tile
constexpr int T128_FRAG = 4 * 4 * 1024;            // a block's fragments: [kc 4][ct 4][lane 64][16 B]
constexpr int T128_BLOCK = T128_FRAG + 1024;       // + tail: fp32 bias [64], factor [64] (beta / gamma; unused for conv4), pad
constexpr int T128_NBLK = 8;

template <typename T>
__global__ __launch_bounds__(512, 2) void naf_tail128_kernel(const NafTail128Params p) {
    __shared__ __attribute__((aligned(16))) char pbuf[2 * T128_BLOCK];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, q = lane & 15, sl = lane >> 4;
    const unsigned pb_lds = (unsigned)(size_t)(lds_ptr_t)pbuf;
    const char* w3b = reinterpret_cast<const char*>(p.w3_scaled);
    const char* wsb = reinterpret_cast<const char*>(p.blocks);

    const long tiles = (p.M + T128_PX - 1) / T128_PX;
    const long t_lo = blockIdx.x * tiles / gridDim.x, t_hi = (long)(blockIdx.x + 1) * tiles / gridDim.x;
    if (t_lo >= t_hi) return;

    auto fetch = [&](int blk, int buf) {             // this wave's pieces of a block (17 KiB: 17 pieces over 8 waves)
        const char* src = blk < 2 ? w3b + (size_t)blk * T128_BLOCK : wsb + (size_t)(blk - 2) * T128_BLOCK;
        for (int i = wave; i < T128_BLOCK / 1024; i += 8) glds16(src + i * 1024, lane * 16, pb_lds + buf * T128_BLOCK + i * 1024);
    };
    fetch(0, 0);
    unsigned g = 0;                                   // blocks done: buffer = g & 1

    for (long tile = t_lo; tile < t_hi; ++tile) {
        const long m0 = tile * T128_PX + 32 * wave + q;
        long mrow[2];
        bool ok[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long m = m0 + 16 * t;
            ok[t] = m < p.M;
            mrow[t] = ok[t] ? m : p.M - 1;
        }
        // x as B fragments (natural K order: 16 bytes = channels 32 kc + 8 s .. + 7); conv3 done, the same registers collect the
        // gated tensor while conv4 still reads nb
        uint4 xb[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                xb[t][kc] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.x) + mrow[t] * p.ldx + 32 * kc + 8 * sl);

        f32x4 y[2][8];
        uint4 nb[2][4];      // LayerNorm2d(y), later the gated tensor, as B fragments in accumulator order
#pragma unroll 1
        for (int blk = 0; blk < T128_NBLK; ++blk, ++g) {
            const char* pb = pbuf + (g & 1) * T128_BLOCK;
            const float* tail = reinterpret_cast<const float*>(pb + T128_FRAG);
            // the block has landed (each wave waits for its own DMAs, the barrier publishes them); every wave is also done with the
            // other buffer, which the next block may now overwrite
            FW_WAIT_VMCNT(0);
            __syncthreads();
            if (blk + 1 < T128_NBLK || tile + 1 < t_hi) fetch(blk + 1 < T128_NBLK ? blk + 1 : 0, (int)((g + 1) & 1));

            const uint4* wl = reinterpret_cast<const uint4*>(pb) + lane;
            f32x4 acc[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (blk < 2) {
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const uint4 wf = wl[(kc * 4 + ct) * 64];
#pragma unroll
                        for (int t = 0; t < 2; ++t) acc[t][ct] = Op<T>::mfma16(wf, xb[t][kc], acc[t][ct]);
                    }
            } else {
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const uint4 wf = wl[(kc * 4 + ct) * 64];
#pragma unroll
                        for (int t = 0; t < 2; ++t) acc[t][ct] = Op<T>::mfma16(wf, nb[t][kc], acc[t][ct]);
                    }
            }

            if (blk < 2) {
                // y = inp + beta * (conv3 + b3), channels 64 blk + 16 ct + 4 s + i
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(tail + 16 * ct + 4 * sl);
                    const f32x4 fa = *reinterpret_cast<const f32x4*>(tail + 64 + 16 * ct + 4 * sl);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f32x4 in = *reinterpret_cast<const f32x4*>(p.stream + mrow[t] * p.lds_ + 64 * blk + 16 * ct + 4 * sl);
                        const f32x4 v = in + (acc[t][ct] + bs) * fa;
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                            if (jj == blk) y[t][4 * jj + ct] = v;
                    }
                }
                if (blk == 1) {
                    // LayerNorm2d over the pixel's 128 channels: 32 of them in this lane, the rest in lanes q + 16, + 32, + 48
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        float s = 0.f;
#pragma unroll
                        for (int ct = 0; ct < 8; ++ct) s += (y[t][ct][0] + y[t][ct][1]) + (y[t][ct][2] + y[t][ct][3]);
                        s += __shfl_xor(s, 16);
                        s += __shfl_xor(s, 32);
                        const float mean = s * (1.0f / 128.0f);
                        float ss = 0.f;
#pragma unroll
                        for (int ct = 0; ct < 8; ++ct) {
                            const f32x4 d = y[t][ct] - mean;
                            ss += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                        }
                        ss += __shfl_xor(ss, 16);
                        ss += __shfl_xor(ss, 32);
                        const float rstd = 1.0f / __builtin_sqrtf(ss * (1.0f / 128.0f) + p.ln_eps);
#pragma unroll
                        for (int kc = 0; kc < 4; ++kc) {
                            const f32x4 a = (y[t][2 * kc] - mean) * rstd, b = (y[t][2 * kc + 1] - mean) * rstd;
                            const uint2 ha = Op<T>::pack4(a[0], a[1], a[2], a[3]), hb = Op<T>::pack4(b[0], b[1], b[2], b[3]);
                            nb[t][kc] = make_uint4(ha.x, ha.y, hb.x, hb.y);
                        }
                    }
                }
            } else if (blk < 6) {
                // conv4 block h = blk - 2: tiles 0, 1 are x1 channels 32 h + 16 ct .., tiles 2, 3 the x2 channels they are gated with
                uint2 hv[2][2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(tail + 16 * ct + 4 * sl);
                    const f32x4 b2 = *reinterpret_cast<const f32x4*>(tail + 32 + 16 * ct + 4 * sl);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f32x4 gt = (acc[t][ct] + b1) * (acc[t][ct + 2] + b2);
                        hv[t][ct] = Op<T>::pack4(gt[0], gt[1], gt[2], gt[3]);
                    }
                }
                // the gated channels 32 h .. 32 h + 31 are conv5's K chunk h; conv4 still reads nb, so they wait in xb
                const int h = blk - 2;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const uint4 v = make_uint4(hv[t][0].x, hv[t][0].y, hv[t][1].x, hv[t][1].y);
#pragma unroll
                    for (int hh = 0; hh < 4; ++hh)
                        if (hh == h) xb[t][hh] = v;
                }
                if (blk == 5) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int kc = 0; kc < 4; ++kc) nb[t][kc] = xb[t][kc];
                }
            } else {
                // out = y + gamma * (conv5 + b5), channels 64 (blk - 6) + 16 ct + 4 s + i
                const int ob = blk - 6;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(tail + 16 * ct + 4 * sl);
                    const f32x4 fa = *reinterpret_cast<const f32x4*>(tail + 64 + 16 * ct + 4 * sl);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        f32x4 yv = y[t][ct];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                            if (jj == ob) yv = y[t][4 * jj + ct];
                        const f32x4 o = yv + (acc[t][ct] + bs) * fa;
                        if (ok[t]) *reinterpret_cast<f32x4*>(p.stream + mrow[t] * p.lds_ + 64 * ob + 16 * ct + 4 * sl) = o;
                    }
                }
            }
        }
    }
}

// w3 blocks with the SCA factor folded in: element e of lane l of fragment (kc, ct) multiplies channel 32 kc + 8 (l >> 4) + e
template <typename T>
__global__ __launch_bounds__(256) void naf_tail128_scale_w3_kernel(const char* __restrict__ src, const float* __restrict__ sca, char* __restrict__ dst) {
    const int total = 2 * T128_BLOCK / 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int blk = i / (T128_BLOCK / 16), r = i - blk * (T128_BLOCK / 16);
        uint4 v = reinterpret_cast<const uint4*>(src)[i];
        if (r < T128_FRAG / 16) {
            const int lane = r & 63, kc = r >> 8;       // r = ((kc * 4 + ct) * 64 + lane)
            const int k0 = 32 * kc + 8 * (lane >> 4);
            const f32x4 lo = Op<T>::unpack4(make_uint2(v.x, v.y)), hi = Op<T>::unpack4(make_uint2(v.z, v.w));
            const uint2 a = Op<T>::pack4(lo[0] * sca[k0], lo[1] * sca[k0 + 1], lo[2] * sca[k0 + 2], lo[3] * sca[k0 + 3]);
            const uint2 b = Op<T>::pack4(hi[0] * sca[k0 + 4], hi[1] * sca[k0 + 5], hi[2] * sca[k0 + 6], hi[3] * sca[k0 + 7]);
            v = make_uint4(a.x, a.y, b.x, b.y);
        }
        reinterpret_cast<uint4*>(dst)[i] = v;
    }
}

size_t naf_tail128_block_bytes() { return (size_t)T128_NBLK * T128_BLOCK; }

void launch_naf_tail128(DType dt, const NafTail128Params& p_in, const float* sca, hipStream_t st) {
    NafTail128Params p = p_in;
    if (!p.x || !p.stream || !p.blocks || !p.w3_scratch || !sca || p.M < 1 || (p.ldx % 8) || (p.lds_ % 4))
        throw Error(1, "naf_tail128: bad argument");
    // conv3's two blocks with this forward's SCA factors
    if (dt == DT_BF16)
        hipLaunchKernelGGL((naf_tail128_scale_w3_kernel<__bf16>), dim3(9), dim3(256), 0, st, (const char*)p.blocks, sca, (char*)p.w3_scratch);
    else
        hipLaunchKernelGGL((naf_tail128_scale_w3_kernel<_Float16>), dim3(9), dim3(256), 0, st, (const char*)p.blocks, sca, (char*)p.w3_scratch);
    p.w3_scaled = p.w3_scratch;
    p.blocks = (const char*)p.blocks + 2 * T128_BLOCK;   // the kernel's blocks 2 .. 7
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long tiles = (p.M + T128_PX - 1) / T128_PX;
    const long grid = tiles < cus ? tiles : cus;
    if (dt == DT_BF16)
        hipLaunchKernelGGL((naf_tail128_kernel<__bf16>), dim3((unsigned)grid), dim3(512), 0, st, p);
    else
        hipLaunchKernelGGL((naf_tail128_kernel<_Float16>), dim3((unsigned)grid), dim3(512), 0, st, p);
    FW_HIP_CHECK(hipGetLastError());
}

// Host-side packer: the eight blocks of a 128-channel NAFBlock's second half (naf_tail128_block_bytes()).
//   blocks 0, 1: conv3 rows 64 b .. 64 b + 63, natural K order (k = 32 kc + 8 (l >> 4) + e), tail = b3, beta
//   blocks 2..5: conv4 with LayerNorm2d's affine part folded in (w' = w ln_w[k], b' = b4 + w ln_b); block h holds rows 32 h .. + 31
//                (tiles 0, 1) and 128 + 32 h .. (tiles 2, 3); accumulator K order (k = 16 (2 kc + (e >> 2)) + 4 (l >> 4) + (e & 3));
//                tail = b' of the 64 rows
//   blocks 6, 7: conv5 rows 64 b .., accumulator K order, tail = b5, gamma
void pack_naf_tail128_blocks(DType dt, const float* w3, const float* b3, const float* beta, const float* ln_w, const float* ln_b, const float* w4,
                             const float* b4, const float* w5, const float* b5, const float* gamma, void* dst_v) {
    constexpr int C = 128;
    char* dst = static_cast<char*>(dst_v);
    memset(dst, 0, naf_tail128_block_bytes());
    for (int blk = 0; blk < T128_NBLK; ++blk) {
        uint16_t* wf = reinterpret_cast<uint16_t*>(dst + (size_t)blk * T128_BLOCK);
        float* tail = reinterpret_cast<float*>(dst + (size_t)blk * T128_BLOCK + T128_FRAG);
        const bool c3 = blk < 2, c4 = blk >= 2 && blk < 6;
        auto row_of = [&](int cc) {   // cc: channel of the block, 0 .. 63
            if (c3) return 64 * blk + cc;
            if (c4) return cc < 32 ? 32 * (blk - 2) + cc : C + 32 * (blk - 2) + (cc - 32);
            return 64 * (blk - 6) + cc;
        };
        const float* w = c3 ? w3 : (c4 ? w4 : w5);
        size_t o = 0;
        for (int kc = 0; kc < 4; ++kc)
            for (int ct = 0; ct < 4; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int row = row_of(16 * ct + (lane & 15));
                        const int k = c3 ? 32 * kc + 8 * (lane >> 4) + e : 16 * (2 * kc + (e >> 2)) + 4 * (lane >> 4) + (e & 3);
                        float v = w[(size_t)row * C + k];
                        if (c4) v *= ln_w[k];
                        wf[o++] = f32_to_operand(dt, v);
                    }
        for (int cc = 0; cc < 64; ++cc) {
            const int row = row_of(cc);
            if (c3) {
                tail[cc] = b3[row];
                tail[64 + cc] = beta[row];
            } else if (c4) {
                double a = b4[row];
                for (int k = 0; k < C; ++k) a += (double)w4[(size_t)row * C + k] * ln_b[k];
                tail[cc] = (float)a;
            } else {
                tail[cc] = b5[row];
                tail[64 + cc] = gamma[row];
            }
        }
    }
}

}  // namespace fw
