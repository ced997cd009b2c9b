// RRDBNet (Real-ESRGAN generator) forward on one MI355X: launch sequencing, workspace and weights.
//
// Architecture: basicsr RRDBNet as constructed by the reference at
// src/framewright/processors/pytorch_realesrgan.py:103-129; block arithmetic as spelled out in-tree at
// src/framewright/processors/aesrgan_face.py:171-204 (ResidualDenseBlock, RRDB) and :249-269 (trunk + tail).
// Data layout: the torch.cat([x, x1, x2, x3, x4]) of a residual dense block is never
// materialised by copying — each RDB owns a 192-channel NHWC "concat" buffer, conv k reads its first
// 64+32(k-1) channels and writes its 32 outputs into the next channel slice.  The residual trunk
// (x5*0.2 + x, and the RRDB-level *0.2 + x) is carried in fp32 side buffers so operand rounding does not
// accumulate over the 69 blocks.
#include <mutex>
#include <vector>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include "fw_internal.h"
#include "../../include/framewright_hip.h"

namespace fw {

struct ConvLayer {
    int cout = 0, cin = 0, ct = 0, chunks = 0;
    void* d_w = nullptr;
    float* d_b = nullptr;
    void* d_wphase = nullptr;   // conv_up1 / conv_up2: the same weights as four 2x2 phase convolutions (conv_up2x_phase.hip)
    void* d_wwino = nullptr;    // conv5 of a dense block, f16: the same weights as row-wise Winograd fragments (conv3x3_wino.hip)
    bool set = false;
};

struct Workspace {
    char* base = nullptr;
    size_t bytes = 0;
};

}  // namespace fw

using namespace fw;

namespace fw {
std::string& last_error_ref() {
    thread_local std::string msg;
    return msg;
}
}  // namespace fw

struct fw_rrdbnet {
    int device = 0;
    fw::StreamOrder order;   // device-side ordering of forwards enqueued on different streams (fw_internal.h)
    int num_block = 0;
    int scale = 4;
    DType dt = DT_BF16;
    std::mutex mu;
    ConvLayer conv_first, conv_body, conv_up1, conv_up2, conv_hr, conv_last;
    std::vector<ConvLayer> body;  // [num_block][3][5]
    Workspace ws;
    int fuse_mask = 3;       // bit 0: conv1+conv2, bit 1: conv3+conv4 (FW_RRDB_FUSE_MASK, for A/B runs)
    bool fuse_pairs = true;  // conv1+conv2 / conv3+conv4 in one kernel (FW_RRDB_FUSE_PAIRS=0 disables, for A/B runs)
    // residual trunk as typed hi + typed lo planes (EPI_RESIDUAL_SPLIT): the residual adds run on the matrix cores as
    // identity chunks and conv5's epilogue loads nothing (FW_RRDB_SPLIT_TRUNK=0 selects the fp32 trunk, for A/B runs)
    bool split_trunk = true;
    // split trunk: lo planes only for the RRDB-level trunk (FW_RRDB_LO=0 keeps a lo plane pair behind every RDB, for A/B runs)
    bool rrdb_lo = true;
    // TIMING-ONLY ablation (wrong pixels): bit 0 - the pair kernels, bit 1 - conv5 read every input chunk from (almost) the same
    // plane, so their HBM reads collapse to about one plane at unchanged MACs: what a launch would cost if its inputs were
    // already on chip (FW_RRDB_ABL_ALIAS; DESIGN.md section 6 uses it to price fusions before building them)
    int abl_alias = 0;
    // conv_up1 / conv_up2 (nearest x2 + 3x3) as four 2x2 phase convolutions on the source grid: 4 instead of 9 taps per output pixel
    // (FW_RRDB_UP_PHASE=0 keeps the gathering nine-tap form, for A/B runs)
    bool up_phase = true;
    // conv5 as a row-wise Winograd F(2, 3): two thirds of the MFMAs (f16 only; conv3x3_wino.hip).  FW_RRDB_C5_WINO: 0 never, 1 (default)
    // rdb1 / rdb2 - no residual planes: 69.4 -> 67.6 ms per 1080p frame -, 2 rdb3 as well (its residual-plane variant spills 17 registers
    // and is 1 ms SLOWER than the direct kernel there: profiles/r03_ab/conv5_winograd_rows.txt)
    int c5_wino = 1;
    bool hr_wino = false;   // conv_hr at the output resolution in the same form (FW_RRDB_HR_WINO=1)
    int abl_rdb3 = 0;   // TIMING-ONLY ablation of rdb3's conv5 (wrong pixels): 1 no lo write, 2 no R lo planes, 4 no R hi planes (FW_RRDB_ABL_RDB3)
    // hipGraph capture of the per-frame forward (BASELINE configs[4] "hipGraph-captured per-frame stages").  graph_mode: 0 never
    // (default), 1 always, 2 for frames of at most graph_max_px input pixels.  Off by default because it buys nothing here: the
    // launches of a forward run back to back at 1080p, and even a 48x64 frame through 3 blocks takes 0.61 ms either way (the
    // persistent 256-workgroup kernels, not the launches, set the floor) - tests/test_rrdbnet_gpu.py prints both.  One executable
    // graph per (frame size, sample bits, buffer pointers), rebuilt when the workspace moves.  FW_RRDB_GRAPH=0|1|2,
    // FW_RRDB_GRAPH_MAX_PX.
    int graph_mode = 0;
    long graph_max_px = 512L * 512L;
    struct GraphEntry {
        int H, W, bits;
        const void* in;
        void* out;
        float* rgb;
        hipGraph_t graph;
        hipGraphExec_t exec;
    };
    std::vector<GraphEntry> graphs;
    bool warmed = false;
    // profiling
    bool profile = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double prof_flops = 0;
    hipStream_t prof_stream = nullptr;
};

namespace {

int fail(int code, const std::string& msg) {
    fw::last_error_ref() = msg;
    return code;
}

template <typename F>
int guarded(F&& f) {
    try {
        f();
        return FW_OK;
    } catch (const fw::Error& e) {
        return fail(e.code, e.what());
    } catch (const std::bad_alloc&) {
        return fail(FW_ERR_OOM, "host out of memory");
    } catch (const std::exception& e) {
        return fail(FW_ERR_INTERNAL, e.what());
    }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Trunk (low-resolution) size for an H x W input.
void trunk_size(const fw_rrdbnet* n, int H, int W, int* Ht, int* Wt) {
    if (n->scale == 2) {
        *Ht = (H + 1) / 2;
        *Wt = (W + 1) / 2;
    } else {
        *Ht = H;
        *Wt = W;
    }
}

struct Plan {
    size_t in_u8, out_u8, in32, cat0, cat1, cat2, F, R, tA, tB, U1, U2, U3, total;
};

Plan make_plan(const fw_rrdbnet* n, int H, int W) {
    int Ht, Wt;
    trunk_size(n, H, W, &Ht, &Wt);
    const size_t px = (size_t)Ht * Wt;
    const size_t s = (size_t)n->scale;
    Plan p{};
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o += align_up(bytes, 256);
        return at;
    };
    p.in_u8 = take((size_t)H * W * 3 * 2);            // staging of host frames: up to 16 bits per sample
    p.out_u8 = take((size_t)H * W * 3 * s * s * 2);
    p.in32 = take(px * 32 * 2);
    // concat buffers: 6 planes of 32 channels (x, x1..x4); the split trunk adds planes 6-7 (lo part of x) and a third
    // buffer, which keeps the RRDB input alive until rdb3's second residual
    const size_t cat_bytes = px * (n->split_trunk ? 256 : 192) * 2;
    p.cat0 = take(cat_bytes);
    p.cat1 = take(cat_bytes);
    const size_t trunk = f32_native_elems(Ht, Wt, 2) * 4;  // fp32 trunk buffers, accumulator-native layout
    p.F = take(trunk);
    if (n->split_trunk) {
        p.cat2 = take(cat_bytes);
    } else {
        p.R = take(trunk);
        p.tA = take(trunk);
        p.tB = take(trunk);
    }
    p.U1 = take(px * 4 * 64 * 2);
    p.U2 = take(px * 16 * 64 * 2);
    p.U3 = take(px * 16 * 64 * 2);
    p.total = o;
    return p;
}

ConvLayer* find_layer(fw_rrdbnet* n, const std::string& key, int* want_cout, int* want_cin) {
    const int in_ch = n->scale == 2 ? 12 : 3;
    auto ret = [&](ConvLayer* l, int co, int ci) {
        *want_cout = co;
        *want_cin = ci;
        return l;
    };
    if (key == "conv_first") return ret(&n->conv_first, 64, in_ch);
    if (key == "conv_body") return ret(&n->conv_body, 64, 64);
    if (key == "conv_up1") return ret(&n->conv_up1, 64, 64);
    if (key == "conv_up2") return ret(&n->conv_up2, 64, 64);
    if (key == "conv_hr") return ret(&n->conv_hr, 64, 64);
    if (key == "conv_last") return ret(&n->conv_last, 3, 64);
    int b = -1, r = -1, c = -1;
    char tail = 0;
    if (sscanf(key.c_str(), "body.%d.rdb%d.conv%d%c", &b, &r, &c, &tail) == 3 && b >= 0 && b < n->num_block && r >= 1 &&
        r <= 3 && c >= 1 && c <= 5) {
        ConvLayer* l = &n->body[((size_t)b * 3 + (r - 1)) * 5 + (c - 1)];
        return ret(l, c == 5 ? 64 : 32, 64 + 32 * (c - 1));
    }
    return nullptr;
}

void free_layer(ConvLayer& l) {
    if (l.d_w) (void)hipFree(l.d_w);
    if (l.d_b) (void)hipFree(l.d_b);
    if (l.d_wphase) (void)hipFree(l.d_wphase);
    if (l.d_wwino) (void)hipFree(l.d_wwino);
    l.d_wphase = nullptr;
    l.d_wwino = nullptr;
    l.d_w = nullptr;
    l.d_b = nullptr;
    l.set = false;
}

double conv_flops(const ConvLayer& l, size_t pixels) { return 2.0 * 9.0 * l.cin * l.cout * (double)pixels; }

void run_conv(fw_rrdbnet* n, const ConvLayer& l, ConvEpilogue epi, ConvParams p, hipStream_t st) {
    p.cin_chunks = l.chunks;
    p.wpk = l.d_w;
    p.bias = l.d_b;
    if (n->profile) {
        if (n->ev_used + 2 > n->ev_pool.size()) {
            size_t old = n->ev_pool.size();
            n->ev_pool.resize(old + 1024);
            for (size_t i = old; i < n->ev_pool.size(); ++i) FW_HIP_CHECK(hipEventCreate(&n->ev_pool[i]));
        }
        FW_HIP_CHECK(hipEventRecord(n->ev_pool[n->ev_used++], st));
        launch_conv3x3(n->dt, l.ct, epi, p, st);
        FW_HIP_CHECK(hipEventRecord(n->ev_pool[n->ev_used++], st));
        n->prof_flops += conv_flops(l, (size_t)p.H * p.W);
        n->prof_stream = st;
    } else {
        launch_conv3x3(n->dt, l.ct, epi, p, st);
    }
}

void run_pair(fw_rrdbnet* n, const ConvLayer& a, const ConvLayer& b, const ConvPairParams& q, hipStream_t st) {
    if (n->profile) {
        if (n->ev_used + 2 > n->ev_pool.size()) {
            size_t old = n->ev_pool.size();
            n->ev_pool.resize(old + 1024);
            for (size_t i = old; i < n->ev_pool.size(); ++i) FW_HIP_CHECK(hipEventCreate(&n->ev_pool[i]));
        }
        FW_HIP_CHECK(hipEventRecord(n->ev_pool[n->ev_used++], st));
        launch_conv3x3_pair(n->dt, q, st);
        FW_HIP_CHECK(hipEventRecord(n->ev_pool[n->ev_used++], st));
        n->prof_flops += conv_flops(a, (size_t)q.H * q.W) + conv_flops(b, (size_t)q.H * q.W);  // algorithmic, no recompute
        n->prof_stream = st;
    } else {
        launch_conv3x3_pair(n->dt, q, st);
    }
}

// bits = 8: d_in / d_out are uint8 BGR; bits = 16: uint16 BGR (range 65535)
void forward(fw_rrdbnet* n, const void* d_in, int bits, int H, int W, void* d_out, float* d_rgb, hipStream_t st) {
    int Ht, Wt;
    trunk_size(n, H, W, &Ht, &Wt);
    const Plan pl = make_plan(n, H, W);
    char* ws = n->ws.base;
    void* in32 = ws + pl.in32;
    void* cat[3] = {ws + pl.cat0, ws + pl.cat1, ws + pl.cat2};
    const int ncat = n->split_trunk ? 3 : 2;

    float* F = (float*)(ws + pl.F);
    float* R = (float*)(ws + pl.R);
    float* tA = (float*)(ws + pl.tA);
    float* tB = (float*)(ws + pl.tB);
    void* U1 = ws + pl.U1;
    void* U2 = ws + pl.U2;
    void* U3 = ws + pl.U3;

    launch_frame_to_nhwc(n->dt, d_in, bits, H, W, in32, 32, n->scale == 2 ? 2 : 1, st);

    // chunk-planar activations: a 192-channel concat buffer is 6 planes of [Ht][Wt][32]
    const long PL = (long)Ht * Wt * 32;
    auto plane = [&](void* buf, int chunk, long pl) { return (void*)((char*)buf + (size_t)chunk * pl * 2); };
    ConvParams base{};
    base.H = Ht;
    base.W = Wt;
    base.s1 = 1.f;
    base.s2 = 1.f;
    base.in_cstride = 32;
    base.in_pstride = PL;
    base.out_cstride = 32;
    base.out_pstride = PL;
    base.f32_native = 1;

    // conv_first -> concat buffer 0 [0:64] + fp32 trunk F          (aesrgan_face.py:250)
    {
        ConvParams p = base;
        p.in = in32;
        p.out = cat[0];
        p.out_f32 = F;
        if (n->split_trunk) p.out_lo = plane(cat[0], 6, PL);
        run_conv(n, n->conv_first, EPI_STORE, p, st);
    }

    int cur = 0;
    for (int b = 0; b < n->num_block; ++b) {
        const float* Rin = (b == 0) ? F : R;
        const int rrdb_in = cur;  // split trunk: planes 0-1 of this buffer stay the RRDB input until rdb3 overwrites them
        for (int k = 0; k < 3; ++k) {
            const ConvLayer* L = &n->body[((size_t)b * 3 + k) * 5];
            // conv1..conv4: growth channels, LeakyReLU(0.2), written into the next plane    (:184-187);
            // fused in pairs (conv1+conv2, conv3+conv4): the shared input chunks leave HBM once per pair
            for (int c = 0; c < 4; c += 2) {
                if (n->fuse_pairs && (n->fuse_mask >> (c >> 1) & 1)) {
                    ConvPairParams q{};
                    q.in = cat[cur];
                    q.in_cstride = 32;
                    q.in_pstride = (n->abl_alias & 1) ? 40 : PL;
                    q.na = 2 + c;
                    q.H = Ht;
                    q.W = Wt;
                    q.wpk_a = L[c].d_w;
                    q.bias_a = L[c].d_b;
                    q.wpk_b = L[c + 1].d_w;
                    q.bias_b = L[c + 1].d_b;
                    q.out_a = plane(cat[cur], 2 + c, PL);
                    q.out_b = plane(cat[cur], 3 + c, PL);
                    q.out_cstride = 32;
                    run_pair(n, L[c], L[c + 1], q, st);
                } else {
                    for (int cc = c; cc < c + 2; ++cc) {
                        ConvParams p = base;
                        p.in = cat[cur];
                        p.out = plane(cat[cur], 2 + cc, PL);
                        p.act = 1;
                        run_conv(n, L[cc], EPI_STORE, p, st);
                    }
                }
            }
            // conv5 + residual(s): x5*0.2 + x  (:188-189); after rdb3 additionally *0.2 + rrdb_in (:204)
            ConvParams p = base;
            const int nxt = (cur + 1) % ncat;
            p.in = cat[cur];
            p.out = cat[nxt];
            p.s1 = 0.2f;
            if (n->abl_alias & 2) p.in_pstride = 40;
            if (n->split_trunk) {
                // y = 0.2*conv5 + x           = 0.2  * (conv5 + 5*x)            x = planes 0,1 (hi) + 6,7 (lo) of cat[cur]
                // y = (0.2*conv5 + x)*0.2 + R = 0.04 * (conv5 + 5*x + 25*R)     R = the same planes of cat[rrdb_in]
                // x hi is added from conv chunks 0,1 while they sit in LDS; residual planes: x lo (6,7) [, R hi (0,1), R lo
                // (6,7)].  Each tile consumes its residual pixels before it stores, so rdb3 updates the RRDB input in place
                // (nxt == rrdb_in).
                const long PB = PL * 2;  // bytes per plane
                const long r_off = (char*)cat[rrdb_in] - (char*)cat[cur];
                p.in_id_scale = 5.f;
                if (n->rrdb_lo) {
                    // lo planes at RRDB granularity: rdb1 / rdb2 carry their output as the typed planes alone (what they lose
                    // - half an ulp of the operand type, once each - reaches the RRDB output scaled by 0.2), and only the
                    // RRDB-level trunk R keeps hi + lo.  rdb1 takes its residual from R hi (the centre tap in LDS), rdb3
                    // adds R hi + R lo and writes hi + lo.  Per RRDB 4 lo-plane transfers instead of 16; measured against
                    // the fp32 oracle the f16 path moves from 3.0-3.7e-4 to 3.4-3.8e-4 max-abs.
                    p.n_id = 0;
                    if (k == 2) {
                        p.out_lo = plane(cat[nxt], 6, PL);
                        p.n_id = 4;
                        p.chunk_off[0] = r_off;              // R hi
                        p.chunk_off[1] = r_off + PB;
                        p.chunk_off[2] = r_off + 6 * PB;     // R lo
                        p.chunk_off[3] = r_off + 7 * PB;
                        for (int i = 0; i < 4; ++i) p.id_scale[i] = 25.f;
                        if (n->abl_rdb3 & 1) {
                            p.out_lo = nullptr;
                        }
                        if (n->abl_rdb3 & 2) p.n_id = 2;
                        if ((n->abl_rdb3 & 6) == 6) p.n_id = 0;
                        else if (n->abl_rdb3 & 4) {
                            p.n_id = 2;
                            p.chunk_off[0] = p.chunk_off[2];
                            p.chunk_off[1] = p.chunk_off[3];
                        }
                    }
                } else {
                    p.out_lo = plane(cat[nxt], 6, PL);
                    p.n_id = (k == 2) ? 6 : 2;
                    p.chunk_off[0] = 6 * PB;
                    p.chunk_off[1] = 7 * PB;
                    p.id_scale[0] = p.id_scale[1] = 5.f;
                    if (k == 2) {
                        p.chunk_off[2] = r_off;              // R hi
                        p.chunk_off[3] = r_off + PB;
                        p.chunk_off[4] = r_off + 6 * PB;     // R lo
                        p.chunk_off[5] = r_off + 7 * PB;
                        for (int i = 2; i < 6; ++i) p.id_scale[i] = 25.f;
                    }
                }
                p.s1 = (k == 2) ? 0.2f * 0.2f : 0.2f;
                if (n->c5_wino && n->dt == DT_F16 && (p.n_id == 0 || n->c5_wino >= 2) && L[4].d_wwino && !(n->abl_alias & 2) && !n->abl_rdb3) {
                    p.cin_chunks = L[4].chunks;
                    p.wpk = L[4].d_wwino;
                    p.bias = L[4].d_b;
                    if (n->profile) {
                        if (n->ev_used + 2 > n->ev_pool.size()) {
                            size_t old = n->ev_pool.size();
                            n->ev_pool.resize(old + 1024);
                            for (size_t i = old; i < n->ev_pool.size(); ++i) FW_HIP_CHECK(hipEventCreate(&n->ev_pool[i]));
                        }
                        FW_HIP_CHECK(hipEventRecord(n->ev_pool[n->ev_used++], st));
                        launch_conv3x3_wino_split(p, st);
                        FW_HIP_CHECK(hipEventRecord(n->ev_pool[n->ev_used++], st));
                        n->prof_flops += conv_flops(L[4], (size_t)p.H * p.W);   // algorithmic: the nine-tap count
                        n->prof_stream = st;
                    } else {
                        launch_conv3x3_wino_split(p, st);
                    }
                } else {
                    run_conv(n, L[4], EPI_RESIDUAL_SPLIT, p, st);
                }
            } else {
                p.res1 = (k == 0) ? Rin : (k == 1 ? tA : tB);
                if (k == 2) {
                    p.res2 = Rin;
                    p.s2 = 0.2f;
                    p.out_f32 = R;
                } else {
                    p.out_f32 = (k == 0) ? tA : tB;
                }
                run_conv(n, L[4], EPI_RESIDUAL, p, st);
            }
            cur = nxt;
        }
    }

    // feat + conv_body(body_feat)                                                  (:256-257)
    {
        ConvParams p = base;
        p.in = cat[cur];
        p.out = cat[(cur + 1) % ncat];
        p.res1 = F;
        p.s1 = 1.f;
        run_conv(n, n->conv_body, EPI_RESIDUAL, p, st);
        cur = (cur + 1) % ncat;
    }
    // lrelu(conv_up1(nearest x2)), lrelu(conv_up2(nearest x2))                    (:260-266)
    auto run_up = [&](const ConvLayer& l, const void* src, long src_pstride, int Hs, int Ws, void* dst, long dst_pstride) {
        if (n->up_phase && l.d_wphase) {
            ConvUpParams u{};
            u.in = src;
            u.in_cstride = 32;
            u.in_pstride = src_pstride;
            u.H = Hs;
            u.W = Ws;
            u.wpk = l.d_wphase;
            u.bias = l.d_b;
            u.act = 1;
            u.out = dst;
            u.out_cstride = 32;
            u.out_pstride = dst_pstride;
            if (n->profile) {
                if (n->ev_used + 2 > n->ev_pool.size()) {
                    size_t old = n->ev_pool.size();
                    n->ev_pool.resize(old + 1024);
                    for (size_t i = old; i < n->ev_pool.size(); ++i) FW_HIP_CHECK(hipEventCreate(&n->ev_pool[i]));
                }
                FW_HIP_CHECK(hipEventRecord(n->ev_pool[n->ev_used++], st));
                launch_conv_up2x_phase(n->dt, u, st);
                FW_HIP_CHECK(hipEventRecord(n->ev_pool[n->ev_used++], st));
                n->prof_flops += conv_flops(l, (size_t)4 * Hs * Ws);   // algorithmic: the nine-tap count of the reference's conv
                n->prof_stream = st;
            } else {
                launch_conv_up2x_phase(n->dt, u, st);
            }
        } else {
            ConvParams p = base;
            p.H = 2 * Hs;
            p.W = 2 * Ws;
            p.in = src;
            p.in_pstride = src_pstride;
            p.upsample2x = 1;
            p.out = dst;
            p.out_pstride = dst_pstride;
            p.act = 1;
            run_conv(n, l, EPI_STORE, p, st);
        }
    };
    run_up(n->conv_up1, cat[cur], PL, Ht, Wt, U1, 4 * PL);
    run_up(n->conv_up2, U1, 4 * PL, 2 * Ht, 2 * Wt, U2, 16 * PL);
    // conv_last(lrelu(conv_hr(feat)))                                              (:268)
    {
        ConvParams p = base;
        p.H = 4 * Ht;
        p.W = 4 * Wt;
        p.in = U2;
        p.in_pstride = 16 * PL;
        p.out = U3;
        p.out_pstride = 16 * PL;
        p.act = 1;
        if (n->c5_wino && n->dt == DT_F16 && n->conv_hr.d_wwino && n->hr_wino) {
            p.cin_chunks = n->conv_hr.chunks;
            p.wpk = n->conv_hr.d_wwino;
            p.bias = n->conv_hr.d_b;
            if (n->profile) {
                if (n->ev_used + 2 > n->ev_pool.size()) {
                    size_t old = n->ev_pool.size();
                    n->ev_pool.resize(old + 1024);
                    for (size_t i = old; i < n->ev_pool.size(); ++i) FW_HIP_CHECK(hipEventCreate(&n->ev_pool[i]));
                }
                FW_HIP_CHECK(hipEventRecord(n->ev_pool[n->ev_used++], st));
                launch_conv3x3_wino_store(p, st);
                FW_HIP_CHECK(hipEventRecord(n->ev_pool[n->ev_used++], st));
                n->prof_flops += conv_flops(n->conv_hr, (size_t)p.H * p.W);
                n->prof_stream = st;
            } else {
                launch_conv3x3_wino_store(p, st);
            }
        } else {
            run_conv(n, n->conv_hr, EPI_STORE, p, st);
        }
    }
    {
        ConvParams p = base;
        p.H = 4 * Ht;
        p.W = 4 * Wt;
        p.in = U3;
        p.in_pstride = 16 * PL;
        if (bits == 16)
            p.out_u16 = (uint16_t*)d_out;
        else
            p.out_u8 = (uint8_t*)d_out;
        p.out_rgb = d_rgb;
        p.img_H = n->scale * H;  // crops the mod-pad of the x2 model
        p.img_W = n->scale * W;
        run_conv(n, n->conv_last, EPI_IMAGE, p, st);
    }
}

}  // namespace

extern "C" {

const char* fw_last_error(void) { return fw::last_error_ref().c_str(); }

int fw_abi_version(void) { return 3; }  // 2: + upscale_u16, resize_lanczos4, grain_addback, attention (softmax rows, transposed pack, MFMA Gram)
                                         // 3 (round 2, additive): + fw_ifnet_*, fw_restormer_*, fw_srvgg_*, fw_unsharp_mask_u8, fw_preserve_edges_*, fw_attn_proj_pack

int fw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fw_rrdbnet_create(int device_id, int num_block, int scale, int dtype, fw_rrdbnet** out) {
    if (!out) return fail(FW_ERR_INVALID, "fw_rrdbnet_create: out is NULL");
    *out = nullptr;
    if (num_block < 1 || num_block > 64) return fail(FW_ERR_INVALID, "fw_rrdbnet_create: num_block out of range");
    if (scale != 2 && scale != 4) return fail(FW_ERR_INVALID, "fw_rrdbnet_create: scale must be 2 or 4");
    if (dtype != FW_DTYPE_BF16 && dtype != FW_DTYPE_F16) return fail(FW_ERR_INVALID, "fw_rrdbnet_create: bad dtype");
    return guarded([&] {
        int ndev = 0;
        FW_HIP_CHECK(hipGetDeviceCount(&ndev));
        if (device_id < 0 || device_id >= ndev) throw Error(FW_ERR_INVALID, "fw_rrdbnet_create: no such device");
        auto n = std::make_unique<fw_rrdbnet>();
        n->device = device_id;
        n->num_block = num_block;
        n->scale = scale;
        n->dt = (DType)dtype;
        n->body.resize((size_t)num_block * 15);
        if (const char* e = getenv("FW_RRDB_FUSE_PAIRS")) n->fuse_pairs = atoi(e) != 0;
        if (const char* e = getenv("FW_RRDB_FUSE_MASK")) n->fuse_mask = atoi(e) & 3;
        if (const char* e = getenv("FW_RRDB_GRAPH")) n->graph_mode = atoi(e);
        if (const char* e = getenv("FW_RRDB_GRAPH_MAX_PX")) n->graph_max_px = atol(e);
        if (const char* e = getenv("FW_RRDB_SPLIT_TRUNK")) n->split_trunk = atoi(e) != 0;
        if (const char* e = getenv("FW_RRDB_LO")) n->rrdb_lo = atoi(e) != 0;
        if (const char* e = getenv("FW_RRDB_ABL_ALIAS")) n->abl_alias = atoi(e);
        if (const char* e = getenv("FW_RRDB_UP_PHASE")) n->up_phase = atoi(e) != 0;
        if (const char* e = getenv("FW_RRDB_ABL_RDB3")) n->abl_rdb3 = atoi(e);
        if (const char* e = getenv("FW_RRDB_C5_WINO")) n->c5_wino = atoi(e);
        if (const char* e = getenv("FW_RRDB_HR_WINO")) n->hr_wino = atoi(e) != 0;
        *out = n.release();
    });
}

int fw_rrdbnet_set_conv(fw_rrdbnet* n, const char* key, const float* weight, const float* bias, int cout, int cin) {
    if (!n || !key || !weight || !bias) return fail(FW_ERR_INVALID, "fw_rrdbnet_set_conv: NULL argument");
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        int wco = 0, wci = 0;
        ConvLayer* l = find_layer(n, key, &wco, &wci);
        if (!l) throw Error(FW_ERR_INVALID, std::string("fw_rrdbnet_set_conv: unknown key '") + key + "'");
        if (cout != wco || cin != wci)
            throw Error(FW_ERR_INVALID, std::string("fw_rrdbnet_set_conv: shape mismatch for '") + key + "': got [" +
                                            std::to_string(cout) + "," + std::to_string(cin) + ",3,3], expected [" +
                                            std::to_string(wco) + "," + std::to_string(wci) + ",3,3]");
        DevGuard dg(n->device);
        free_layer(*l);
        l->cout = cout;
        l->cin = cin;
        l->ct = (cout + 31) / 32;
        l->chunks = (cin + 31) / 32;
        const size_t ne = pack_conv3x3_weights(n->dt, nullptr, cout, cin, l->ct, l->chunks, nullptr);
        std::vector<uint16_t> packed(ne);
        pack_conv3x3_weights(n->dt, weight, cout, cin, l->ct, l->chunks, packed.data());
        std::vector<float> b(32 * l->ct, 0.f);
        for (int i = 0; i < cout; ++i) b[i] = bias[i];
        FW_HIP_CHECK(hipMalloc(&l->d_w, ne * 2));
        FW_HIP_CHECK(hipMalloc((void**)&l->d_b, b.size() * 4));
        FW_HIP_CHECK(hipMemcpy(l->d_w, packed.data(), ne * 2, hipMemcpyHostToDevice));
        FW_HIP_CHECK(hipMemcpy(l->d_b, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        if (n->dt == DT_F16 && cout == 64 && (cin == 192 || l == &n->conv_hr)) {   // a dense block's conv5; conv_hr of the tail
            const size_t nw = pack_conv3x3_wino_weights(n->dt, nullptr, cout, cin, l->chunks, nullptr);
            std::vector<uint16_t> wn(nw);
            pack_conv3x3_wino_weights(n->dt, weight, cout, cin, l->chunks, wn.data());
            FW_HIP_CHECK(hipMalloc(&l->d_wwino, nw * 2));
            FW_HIP_CHECK(hipMemcpy(l->d_wwino, wn.data(), nw * 2, hipMemcpyHostToDevice));
        }
        if (l == &n->conv_up1 || l == &n->conv_up2) {
            const size_t np = pack_conv_up2x_phase_weights(n->dt, nullptr, nullptr);
            std::vector<uint16_t> ph(np);
            pack_conv_up2x_phase_weights(n->dt, weight, ph.data());
            FW_HIP_CHECK(hipMalloc(&l->d_wphase, np * 2));
            FW_HIP_CHECK(hipMemcpy(l->d_wphase, ph.data(), np * 2, hipMemcpyHostToDevice));
        }
        l->set = true;
    });
}

int fw_rrdbnet_finalize(fw_rrdbnet* n) {
    if (!n) return fail(FW_ERR_INVALID, "fw_rrdbnet_finalize: NULL");
    std::lock_guard<std::mutex> lk(n->mu);
    const ConvLayer* singles[] = {&n->conv_first, &n->conv_body, &n->conv_up1, &n->conv_up2, &n->conv_hr, &n->conv_last};
    const char* names[] = {"conv_first", "conv_body", "conv_up1", "conv_up2", "conv_hr", "conv_last"};
    for (int i = 0; i < 6; ++i)
        if (!singles[i]->set) return fail(FW_ERR_INVALID, std::string("fw_rrdbnet_finalize: missing ") + names[i]);
    for (size_t i = 0; i < n->body.size(); ++i)
        if (!n->body[i].set)
            return fail(FW_ERR_INVALID, "fw_rrdbnet_finalize: missing body." + std::to_string(i / 15) + ".rdb" +
                                            std::to_string((i / 5) % 3 + 1) + ".conv" + std::to_string(i % 5 + 1));
    return FW_OK;
}

size_t fw_rrdbnet_workspace_bytes(const fw_rrdbnet* n, int H, int W) {
    if (!n || H < 1 || W < 1) return 0;
    return make_plan(n, H, W).total;
}

double fw_rrdbnet_flops(const fw_rrdbnet* n, int H, int W) {
    if (!n || H < 1 || W < 1) return 0.0;
    int Ht, Wt;
    trunk_size(n, H, W, &Ht, &Wt);
    const double px = (double)Ht * Wt;
    const double in_ch = n->scale == 2 ? 12 : 3;
    double mac = 9.0 * in_ch * 64;                                             // conv_first
    mac += n->num_block * 3.0 * 9.0 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64);  // RRDB trunk
    mac += 9.0 * 64 * 64;                                                      // conv_body
    mac += 9.0 * 64 * 64 * 4;                                                  // conv_up1 at 2x
    mac += 9.0 * 64 * 64 * 16 * 2;                                             // conv_up2, conv_hr at 4x
    mac += 9.0 * 64 * 3 * 16;                                                  // conv_last
    return 2.0 * mac * px;
}

static void drop_graphs(fw_rrdbnet* n) {
    for (auto& g : n->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    n->graphs.clear();
}

static int upscale_any(fw_rrdbnet* n, const void* in_bgr, int in_loc, int bits, int H, int W, void* out_bgr, int out_loc,
                       float* out_rgb_f32, void* stream, const char* who) {
    const std::string w(who);
    if (!n || !in_bgr) return fail(FW_ERR_INVALID, w + ": NULL argument");
    if (!out_bgr && !out_rgb_f32) return fail(FW_ERR_INVALID, w + ": no output requested");
    if (H < 1 || W < 1 || H > 16384 || W > 16384) return fail(FW_ERR_INVALID, w + ": bad frame size");
    if (n->scale == 2 && (H < 2 || W < 2)) return fail(FW_ERR_INVALID, w + ": x2 needs >= 2x2 input");
    if ((in_loc != FW_HOST && in_loc != FW_DEVICE) || (out_loc != FW_HOST && out_loc != FW_DEVICE))
        return fail(FW_ERR_INVALID, w + ": bad buffer location");
    int rc = fw_rrdbnet_finalize(n);
    if (rc != FW_OK) return rc;
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        DevGuard dg(n->device);
        hipStream_t st = (hipStream_t)stream;
        StreamOrder::Scope in_order(n->order, st);
        const Plan pl = make_plan(n, H, W);
        if (n->ws.bytes < pl.total) {
            // the previous workspace may still be in use by work queued on some stream
            FW_HIP_CHECK(hipDeviceSynchronize());
            drop_graphs(n);   // they hold the old workspace's addresses
            if (n->ws.base) (void)hipFree(n->ws.base);
            n->ws.base = nullptr;
            n->ws.bytes = 0;
            FW_HIP_CHECK(hipMalloc((void**)&n->ws.base, pl.total));
            n->ws.bytes = pl.total;
        }
        const size_t in_bytes = (size_t)H * W * 3 * (bits / 8);
        const size_t out_bytes = in_bytes * n->scale * n->scale;
        const void* d_in = in_bgr;
        if (in_loc == FW_HOST) {
            void* stage = n->ws.base + pl.in_u8;
            FW_HIP_CHECK(hipMemcpyAsync(stage, in_bgr, in_bytes, hipMemcpyHostToDevice, st));
            d_in = stage;
        }
        void* d_out = out_bgr;
        if (out_bgr && out_loc == FW_HOST) d_out = n->ws.base + pl.out_u8;
        // (the engine's first forward always runs uncaptured: one-time initialisations inside the launchers must not land in a
        // capture)
        const bool graphed = n->warmed && !n->profile &&
                             (n->graph_mode == 1 || (n->graph_mode == 2 && (long)H * W <= n->graph_max_px));
        n->warmed = true;
        if (!graphed) {
            forward(n, d_in, bits, H, W, d_out, out_rgb_f32, st);
        } else {
            fw_rrdbnet::GraphEntry* hit = nullptr;
            for (auto& g : n->graphs)
                if (g.H == H && g.W == W && g.bits == bits && g.in == d_in && g.out == d_out && g.rgb == out_rgb_f32) hit = &g;
            if (!hit) {
                if (n->graphs.size() >= 16) drop_graphs(n);   // callers that never reuse their buffers: do not grow without bound
                (void)conv_zero_page();                       // its first use allocates: not inside a capture
                // capture on a stream of our own: the caller's stream may be the legacy default stream, which cannot capture
                hipStream_t cs = nullptr;
                FW_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
                fw_rrdbnet::GraphEntry e{H, W, bits, d_in, d_out, out_rgb_f32, nullptr, nullptr};
                hipError_t err = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
                if (err == hipSuccess) {
                    try {
                        forward(n, d_in, bits, H, W, d_out, out_rgb_f32, cs);
                    } catch (...) {
                        hipGraph_t junk = nullptr;
                        (void)hipStreamEndCapture(cs, &junk);
                        if (junk) (void)hipGraphDestroy(junk);
                        (void)hipStreamDestroy(cs);
                        throw;
                    }
                    err = hipStreamEndCapture(cs, &e.graph);
                }
                if (err == hipSuccess) err = hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0);
                (void)hipStreamDestroy(cs);
                if (err != hipSuccess) {
                    if (e.graph) (void)hipGraphDestroy(e.graph);
                    FW_HIP_CHECK(err);
                }
                n->graphs.push_back(e);
                hit = &n->graphs.back();
            }
            FW_HIP_CHECK(hipGraphLaunch(hit->exec, st));
        }
        if (out_bgr && out_loc == FW_HOST) {
            FW_HIP_CHECK(hipMemcpyAsync(out_bgr, d_out, out_bytes, hipMemcpyDeviceToHost, st));
            FW_HIP_CHECK(hipStreamSynchronize(st));
        }
    });
}

int fw_rrdbnet_upscale_u8(fw_rrdbnet* n, const uint8_t* in_bgr, int in_loc, int H, int W, uint8_t* out_bgr,
                          int out_loc, float* out_rgb_f32, void* stream) {
    return upscale_any(n, in_bgr, in_loc, 8, H, W, out_bgr, out_loc, out_rgb_f32, stream, "fw_rrdbnet_upscale_u8");
}

int fw_rrdbnet_upscale_u16(fw_rrdbnet* n, const uint16_t* in_bgr, int in_loc, int H, int W, uint16_t* out_bgr,
                           int out_loc, float* out_rgb_f32, void* stream) {
    return upscale_any(n, in_bgr, in_loc, 16, H, W, out_bgr, out_loc, out_rgb_f32, stream, "fw_rrdbnet_upscale_u16");
}

int fw_rrdbnet_profile_enable(fw_rrdbnet* n, int on) {
    if (!n) return fail(FW_ERR_INVALID, "fw_rrdbnet_profile_enable: NULL");
    std::lock_guard<std::mutex> lk(n->mu);
    n->profile = on != 0;
    n->ev_used = 0;
    n->prof_flops = 0;
    return FW_OK;
}

int fw_rrdbnet_profile_read(fw_rrdbnet* n, int* launches, double* total_ms, double* total_flops) {
    if (!n) return fail(FW_ERR_INVALID, "fw_rrdbnet_profile_read: NULL");
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        DevGuard dg(n->device);
        double ms = 0;
        if (n->ev_used) FW_HIP_CHECK(hipEventSynchronize(n->ev_pool[n->ev_used - 1]));
        for (size_t i = 0; i + 1 < n->ev_used; i += 2) {
            float t = 0;
            FW_HIP_CHECK(hipEventElapsedTime(&t, n->ev_pool[i], n->ev_pool[i + 1]));
            ms += t;
        }
        if (launches) *launches = (int)(n->ev_used / 2);
        if (total_ms) *total_ms = ms;
        if (total_flops) *total_flops = n->prof_flops;
        n->ev_used = 0;
        n->prof_flops = 0;
    });
}

int fw_rrdbnet_destroy(fw_rrdbnet* n) {
    if (!n) return FW_OK;
    // a call in flight on another thread finishes first; calling INTO a destroyed handle remains the caller's bug (the Python
    // engine counts its calls in flight and destroys only when there are none)
    { std::lock_guard<std::mutex> lk(n->mu); }
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(n->device);
    (void)hipDeviceSynchronize();
    for (auto& l : n->body) free_layer(l);
    free_layer(n->conv_first);
    free_layer(n->conv_body);
    free_layer(n->conv_up1);
    free_layer(n->conv_up2);
    free_layer(n->conv_hr);
    free_layer(n->conv_last);
    drop_graphs(n);
    if (n->ws.base) (void)hipFree(n->ws.base);
    for (auto e : n->ev_pool) (void)hipEventDestroy(e);
    if (prev >= 0) (void)hipSetDevice(prev);
    n->order.destroy();
    delete n;
    return FW_OK;
}

size_t fw_pack_conv3x3(int dtype, const float* weight, int cout, int cin, int cout_tiles, int cin_chunks,
                       uint16_t* dst) {
    if (cout < 1 || cin < 1 || cout_tiles < 1 || cin_chunks < 1 || cout > 32 * cout_tiles || cin > 32 * cin_chunks)
        return 0;
    if (dst && !weight) return 0;
    return pack_conv3x3_weights((DType)dtype, weight, cout, cin, cout_tiles, cin_chunks, dst);
}

int fw_conv3x3_nhwc(int dtype, const void* x, int in_cstride, long in_plane_stride, int cin_chunks, int H, int W,
                    const void* packed_weight, const float* bias, int cout_tiles, int act_lrelu, int upsample2x,
                    const float* res1, float s1, const float* res2, float s2, void* out, int out_cstride,
                    long out_plane_stride, int out_coff, float* out_f32, void* stream) {
    return fw_conv3x3_nhwc_ex(dtype, x, in_cstride, in_plane_stride, cin_chunks, H, W, packed_weight, bias, cout_tiles,
                              act_lrelu, upsample2x, res1, s1, res2, s2, nullptr, 0, 0, 0, out, out_cstride,
                              out_plane_stride, out_coff, out_f32, stream);
}

int fw_conv3x3_pair_nhwc(int dtype, const void* x, int in_cstride, long in_plane_stride, int in_chunks, int H, int W,
                         const void* packed_weight_a, const float* bias_a, const void* packed_weight_b, const float* bias_b,
                         void* out_a, void* out_b, int out_cstride, void* stream) {
    if (dtype != FW_DTYPE_BF16 && dtype != FW_DTYPE_F16) return fail(FW_ERR_INVALID, "fw_conv3x3_pair_nhwc: bad dtype");
    return guarded([&] {
        ConvPairParams q{};
        q.in = x;
        q.in_cstride = in_cstride;
        q.in_pstride = in_plane_stride > 0 ? in_plane_stride : 32;
        q.na = in_chunks;
        q.H = H;
        q.W = W;
        q.wpk_a = packed_weight_a;
        q.bias_a = bias_a;
        q.wpk_b = packed_weight_b;
        q.bias_b = bias_b;
        q.out_a = out_a;
        q.out_b = out_b;
        q.out_cstride = out_cstride;
        if (q.in_pstride == 32 && in_cstride < 32 * in_chunks) throw Error(FW_ERR_INVALID, "fw_conv3x3_pair_nhwc: input too narrow");
        launch_conv3x3_pair((DType)dtype, q, (hipStream_t)stream);
    });
}

int fw_conv3x3_nhwc_ex(int dtype, const void* x, int in_cstride, long in_plane_stride, int cin_chunks, int H, int W,
                       const void* packed_weight, const float* bias, int cout_tiles, int act_lrelu, int upsample2x,
                       const float* res1, float s1, const float* res2, float s2, const float* chan_scale, int post_act,
                       int f32_cstride, int f32_coff, void* out, int out_cstride, long out_plane_stride, int out_coff,
                       float* out_f32, void* stream) {
    if (!x || !packed_weight || !bias) return fail(FW_ERR_INVALID, "fw_conv3x3_nhwc: NULL argument");
    if (dtype != FW_DTYPE_BF16 && dtype != FW_DTYPE_F16) return fail(FW_ERR_INVALID, "fw_conv3x3_nhwc: bad dtype");
    if (cout_tiles != 1 && cout_tiles != 2) return fail(FW_ERR_INVALID, "fw_conv3x3_nhwc: cout_tiles must be 1 or 2");
    if (res1 && cout_tiles != 2) return fail(FW_ERR_INVALID, "fw_conv3x3_nhwc: residual epilogue needs 64 channels");
    if (!res1 && res2) return fail(FW_ERR_INVALID, "fw_conv3x3_nhwc: res2 without res1");
    if (act_lrelu == 2 && (res1 || !chan_scale)) return fail(FW_ERR_INVALID, "fw_conv3x3_nhwc: PReLU needs slopes and no residual");
    return guarded([&] {
        ConvParams p{};
        p.in = x;
        p.in_cstride = in_cstride;
        p.in_pstride = in_plane_stride > 0 ? in_plane_stride : 32;
        p.out_pstride = out_plane_stride > 0 ? out_plane_stride : 32;
        p.cin_chunks = cin_chunks;
        p.H = H;
        p.W = W;
        p.wpk = packed_weight;
        p.bias = bias;
        p.out = out;
        p.out_cstride = out_cstride;
        p.out_coff = out_coff;
        p.out_f32 = out_f32;
        p.res1 = res1;
        p.res2 = res2;
        p.s1 = s1;
        p.s2 = s2;
        p.act = act_lrelu;
        p.upsample2x = upsample2x;
        p.chan_scale = chan_scale;
        p.post_act = post_act;
        p.f32_cstride = f32_cstride;
        p.f32_coff = f32_coff;
        launch_conv3x3((DType)dtype, cout_tiles, res1 ? EPI_RESIDUAL : EPI_STORE, p, (hipStream_t)stream);
    });
}

size_t fw_pack_conv_up2x_phase(int dtype, const float* weight, uint16_t* dst) {
    if (dtype != FW_DTYPE_BF16 && dtype != FW_DTYPE_F16) return 0;
    if (dst && !weight) return 0;
    return pack_conv_up2x_phase_weights((DType)dtype, weight, dst);
}

int fw_conv_up2x_phase_nhwc(int dtype, const void* x, int in_cstride, long in_plane_stride, int H, int W, const void* packed_weight,
                            const float* bias, int act_lrelu, void* out, int out_cstride, long out_plane_stride, void* stream) {
    if (!x || !packed_weight || !bias || !out) return fail(FW_ERR_INVALID, "fw_conv_up2x_phase_nhwc: NULL argument");
    if (dtype != FW_DTYPE_BF16 && dtype != FW_DTYPE_F16) return fail(FW_ERR_INVALID, "fw_conv_up2x_phase_nhwc: bad dtype");
    if (H < 1 || W < 1) return fail(FW_ERR_INVALID, "fw_conv_up2x_phase_nhwc: bad size");
    return guarded([&] {
        ConvUpParams u{};
        u.in = x;
        u.in_cstride = in_cstride;
        u.in_pstride = in_plane_stride > 0 ? in_plane_stride : 32;
        u.H = H;
        u.W = W;
        u.wpk = packed_weight;
        u.bias = bias;
        u.act = act_lrelu;
        u.out = out;
        u.out_cstride = out_cstride;
        u.out_pstride = out_plane_stride > 0 ? out_plane_stride : 32;
        launch_conv_up2x_phase((DType)dtype, u, (hipStream_t)stream);
    });
}

int fw_u8_to_nhwc(int dtype, const uint8_t* in_bgr, int H, int W, void* out, int out_cstride, void* stream) {
    if (!in_bgr || !out || H < 1 || W < 1) return fail(FW_ERR_INVALID, "fw_u8_to_nhwc: bad argument");
    if (dtype != FW_DTYPE_BF16 && dtype != FW_DTYPE_F16) return fail(FW_ERR_INVALID, "fw_u8_to_nhwc: bad dtype");
    return guarded([&] { launch_u8_to_nhwc((DType)dtype, in_bgr, H, W, out, out_cstride, 1, (hipStream_t)stream); });
}

int fw_pixel_shuffle_add_u8(const float* conv, int conv_cstride, const uint8_t* in_bgr, int H, int W, int scale,
                            uint8_t* out_bgr, float* out_rgb_f32, void* stream) {
    if (!conv || !in_bgr || (!out_bgr && !out_rgb_f32) || H < 1 || W < 1)
        return fail(FW_ERR_INVALID, "fw_pixel_shuffle_add_u8: bad argument");
    return guarded([&] {
        launch_pixel_shuffle_add(conv, conv_cstride, in_bgr, H, W, scale, out_bgr, out_rgb_f32, (hipStream_t)stream);
    });
}

}  // extern "C"
