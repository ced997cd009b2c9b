// IFNet v4.6 (RIFE) as ONE engine behind the C-ABI: weights, workspace, launch sequencing, optional hipGraph replay.
//
// What it replaces: the reference's `rife-ncnn-vulkan -i in -o out -m rife-v4.6 ...` subprocess per pass
// (src/framewright/processors/interpolation.py:628-650; SURVEY.md section 8(b) lists `fw_interp_u8(h, f0, f1, t, out)` as the
// entry a binder of this library needs).  Round 1 sequenced the ~100 launches of a forward from Python over the building
// blocks of ifnet_ops.hip; here the same launches are issued from C++ under one mutex per handle, so a non-Python binder
// gets the operator, not the bricks, and the launch-bound low-resolution part of the net can be captured in a hipGraph
// (BASELINE configs[4]: "hipGraph-captured per-frame stages").
//
// Arithmetic: IFNet_HDv3 v4.6 as recorded in SURVEY.md section A.5 and restated in oracle/ifnet_ref.py (parity vs the
// upstream binary unpinned).  Four IFBlocks at scales 8, 4, 2, 1 with 192, 128, 96, 64 channels:
//   conv0    two stride-2 3x3 convs (+LeakyReLU 0.2)  -> 3x3 stride-1 convs on pixel-unshuffled tensors (MFMA conv kernel)
//   8 x ResConv   lrelu(conv3x3(x) * beta + x)        -> MFMA conv kernel with the residual / per-channel-scale epilogue
//   lastconv ConvTranspose2d(c, 24, 4, 2, 1) + PixelShuffle(2) -> one 3x3 conv with 96 outputs + depth-to-space(4)
// resize / backward warp / mask blend: the HBM-bound kernels of ifnet_ops.hip.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "fw_internal.h"
#include "../../include/framewright_hip.h"

using namespace fw;

namespace {

constexpr int NBLK = 4;
constexpr int CH[NBLK] = {192, 128, 96, 64};
constexpr int SC[NBLK] = {8, 4, 2, 1};
constexpr int NRES = 8;

int pad_to(int n, int m) { return (n + m - 1) / m * m; }

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

void upload(DevBuf& b, const void* src, size_t bytes) {
    b.release();
    FW_HIP_CHECK(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    FW_HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
}

// One 3x3 convolution [cout_pad][cin_pad] split into launches of 64 (or a last 32) output channels.
struct Conv {
    int cin_pad = 0, cout_pad = 0;
    int cin_chunks = 0;      // 32-channel chunks that hold real input channels: the K loop stops there (a 96-channel block pads its tensors to 128;
                             // the fourth chunk of every input is zeros times zero weights)
    struct Group {
        DevBuf w, b;
        int ct = 0, off = 0;
    };
    std::vector<Group> groups;
    DevBuf wall, ball;       // every group's weights / biases in one buffer each, when all groups are 64 channels wide:
    size_t gstride = 0;      // the groups then run side by side in one launch (ConvParams::n_groups)
    DevBuf nwall;            // the same conv as 32-channel groups (one launch, ball serves both): twice the workgroups with half the
    size_t ngstride = 0;     // MFMAs each, for maps of a few tiles (build(..., narrow = true))
    void release() {
        for (auto& g : groups) {
            g.w.release();
            g.b.release();
        }
        groups.clear();
        wall.release();
        ball.release();
        nwall.release();
        gstride = ngstride = 0;
    }
    // w: [cout][cin][3][3] fp32 (already transformed), zero-padded to [cout_pad][cin_pad]
    void build(DType dt, const std::vector<float>& w, const std::vector<float>& b, int cout, int cin, int cin_p, int cout_p, bool narrow = false) {
        release();
        cin_pad = cin_p;
        cout_pad = cout_p;
        cin_chunks = (cin + 31) / 32;
        std::vector<float> wp((size_t)cout_p * cin_p * 9, 0.f), bp(cout_p, 0.f);
        for (int co = 0; co < cout; ++co) {
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < 9; ++t) wp[((size_t)co * cin_p + ci) * 9 + t] = w[((size_t)co * cin + ci) * 9 + t];
            bp[co] = b[co];
        }
        for (int off = 0; off < cout_p;) {
            const int ct = cout_p - off >= 64 ? 2 : 1;
            const int chunks = cin_p / 32;
            groups.emplace_back();
            Group& g = groups.back();
            g.ct = ct;
            g.off = off;
            std::vector<uint16_t> pk(pack_conv3x3_weights(dt, nullptr, 32 * ct, cin_p, ct, chunks, nullptr));
            pack_conv3x3_weights(dt, wp.data() + (size_t)off * cin_p * 9, 32 * ct, cin_p, ct, chunks, pk.data());
            upload(g.w, pk.data(), pk.size() * 2);
            upload(g.b, bp.data() + off, (size_t)32 * ct * 4);
            off += 32 * ct;
        }
        if (groups.size() > 1 && cout_p % 64 == 0) {
            const int chunks = cin_p / 32;
            const size_t n = pack_conv3x3_weights(dt, nullptr, 64, cin_p, 2, chunks, nullptr);
            std::vector<uint16_t> all(n * groups.size());
            for (size_t g = 0; g < groups.size(); ++g)
                pack_conv3x3_weights(dt, wp.data() + g * 64 * cin_p * 9, 64, cin_p, 2, chunks, all.data() + g * n);
            upload(wall, all.data(), all.size() * 2);
            upload(ball, bp.data(), bp.size() * 4);
            gstride = n * 2;
            if (narrow) {
                const size_t n1 = pack_conv3x3_weights(dt, nullptr, 32, cin_p, 1, chunks, nullptr);
                std::vector<uint16_t> all1(n1 * (cout_p / 32));
                for (int g = 0; g < cout_p / 32; ++g)
                    pack_conv3x3_weights(dt, wp.data() + (size_t)g * 32 * cin_p * 9, 32, cin_p, 1, chunks, all1.data() + g * n1);
                upload(nwall, all1.data(), all1.size() * 2);
                ngstride = n1 * 2;
            }
        }
    }
};

// Conv2d(cin, cout, 3, stride 2, pad 1) == 3x3 / s1 / p1 conv on pixel_unshuffle(x, 2) with
// W'[co][ci * 4 + dy * 2 + dx][U][V]: tap ky -> (U, dy) = {0: (0, 1), 1: (1, 0), 2: (1, 1)}; the taps with U = 2 or V = 2 are zero
std::vector<float> stride2_as_unshuffled(const float* w, int cout, int cin) {
    std::vector<float> out((size_t)cout * cin * 4 * 9, 0.f);
    const int mU[3] = {0, 1, 1}, mD[3] = {1, 0, 1};
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx)
                    out[((size_t)co * cin * 4 + ci * 4 + mD[ky] * 2 + mD[kx]) * 9 + mU[ky] * 3 + mU[kx]] = w[((size_t)co * cin + ci) * 9 + ky * 3 + kx];
    return out;
}

// ConvTranspose2d(cin, cout, 4, stride 2, pad 1) == 3x3 conv producing cout * 4 channels n = co * 4 + py * 2 + px (the output
// parity): tap dy (input row y + dy) uses kernel row ky = py + 1 - 2 dy when 0 <= ky <= 3.  w: [cin][cout][4][4].
void convtranspose_as_3x3(const float* w, const float* b, int cin, int cout, std::vector<float>* w3, std::vector<float>* b3) {
    w3->assign((size_t)cout * 4 * cin * 9, 0.f);
    b3->assign((size_t)cout * 4, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int par = 0; par < 4; ++par) {
            const int py = par >> 1, px = par & 1;
            (*b3)[co * 4 + par] = b[co];
            for (int dy = -1; dy <= 1; ++dy) {
                const int ky = py + 1 - 2 * dy;
                if (ky < 0 || ky > 3) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    const int kx = px + 1 - 2 * dx;
                    if (kx < 0 || kx > 3) continue;
                    for (int ci = 0; ci < cin; ++ci)
                        (*w3)[((size_t)(co * 4 + par) * cin + ci) * 9 + (dy + 1) * 3 + (dx + 1)] = w[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx];
                }
            }
        }
}

struct Block {
    int c = 0, cin = 0, c2p = 0, cp = 0;
    std::vector<float> h_w00, h_b00, h_w01, h_b01, h_wl, h_bl;   // host copies until finalize()
    std::vector<float> h_wr[NRES], h_br[NRES], h_beta[NRES];
    Conv conv00, conv01, res[NRES], last;
    Conv res_split[NRES];   // cp == 64: the ResConvs with beta folded into weights and bias (split trunk: fw_ifnet::split_trunk)
    DevBuf beta[NRES];
    unsigned have = 0;   // bits: 0 w00, 1 b00, 2 w01, 3 b01, 4 wl, 5 bl, 6 + 3j (w), 7 + 3j (b), 8 + 3j (beta)
};
constexpr unsigned BLOCK_ALL = (1u << (6 + 3 * NRES)) - 1;

struct Plan {
    size_t in0, in1, out_u8, I0, I1, flow, mask, X, xin, u0, a0, u1, featA, featB, f32A, f32B, t96, tmp, total;
};

}  // namespace

struct fw_ifnet {
    int device = 0;
    fw::StreamOrder order;   // device-side ordering of forwards enqueued on different streams (fw_internal.h)
    DType dt = DT_F16;
    std::mutex mu;
    Block blk[NBLK];
    bool built = false;
    DevBuf ws;
    // hipGraph replay of a forward, keyed by everything a captured launch sequence bakes in (FW_IFNET_GRAPH=1; off by default:
    // a caller that hands over fresh buffers every frame would re-capture every frame)
    int graph_mode = 0;
    bool merge_groups = true;   // the 64-channel output groups of a conv in one launch (FW_IFNET_MERGE_GROUPS=0: A/B)
    bool fuse_glue = true;      // an IFBlock's input in one kernel, depth-to-space inside the accumulate (FW_IFNET_FUSE_GLUE=0: A/B)
    bool native_trunk = true;   // the ResConv chain's fp32 trunk in the conv kernel's accumulator-native layout (FW_IFNET_NATIVE_TRUNK=0: A/B)
    // A 64-channel IFBlock (the full-resolution one: eight ResConvs on a 270 x 480 map at 1080p, HBM-bound: 33 MB of fp32 trunk in and
    // out per conv beside the typed tensor) keeps its trunk as two operand-typed tensors, hi = T(x) and lo = T(x - hi), like the RRDBNet
    // (DESIGN.md section 2): hi is the tensor the next conv reads anyway and comes back as a residual from the tile already in LDS (an
    // identity MFMA), lo is the only extra read; lrelu(conv(x) * beta + x) with beta folded into the conv's weights and bias
    // (EPI_RESIDUAL_SPLIT).  68 instead of 100 MB per ResConv.  FW_IFNET_SPLIT_TRUNK=0: the fp32 trunk (A/B).
    bool split_trunk = true;
    bool skip_pad_chunks = true;   // FW_IFNET_SKIP_PAD=0: walk the zero-padded input chunks as well (A/B)
    bool narrow_groups = true;  // 32-channel output groups for the conv chains of blocks with few tiles (FW_IFNET_NARROW=0: A/B;
    long narrow_below = 128;    //   FW_IFNET_NARROW_BELOW: below that many 64-channel workgroups per launch)
    bool warmed = false;
    struct GraphEntry {
        int H, W;
        float t;
        const void *a, *b;
        void *out, *rgb;
        hipGraph_t graph;
        hipGraphExec_t exec;
    };
    std::vector<GraphEntry> graphs;
};

namespace {

int fail(int code, const std::string& m) {
    fw::last_error_ref() = m;
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

size_t esz(DType) { return 2; }

Plan make_plan(int H, int W) {
    const size_t Hp = pad_to(H, 32), Wp = pad_to(W, 32);
    Plan p{};
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += (bytes + 255) / 256 * 256;
        return at;
    };
    p.in0 = take((size_t)H * W * 3);
    p.in1 = take((size_t)H * W * 3);
    p.out_u8 = take((size_t)H * W * 3);
    p.I0 = take(Hp * Wp * 3 * 4);
    p.I1 = take(Hp * Wp * 3 * 4);
    p.flow = take(Hp * Wp * 4 * 4);
    p.mask = take(Hp * Wp * 4);
    p.X = take(Hp * Wp * 8 * 4);
    size_t xin = 0, u0 = 0, a0 = 0, u1 = 0, feat = 0, f32 = 0, t96 = 0, tmp = 0;
    for (int i = 0; i < NBLK; ++i) {
        const size_t hs = Hp / SC[i], ws = Wp / SC[i], cin = i == 0 ? 7 : 12;
        const size_t c2p = pad_to(CH[i] / 2, 32), cp = pad_to(CH[i], 64);
        xin = std::max(xin, hs * ws * cin * 4);
        u0 = std::max(u0, hs / 2 * (ws / 2) * pad_to(4 * (int)cin, 32) * 2);
        a0 = std::max(a0, hs / 2 * (ws / 2) * c2p * 2);
        u1 = std::max(u1, hs / 4 * (ws / 4) * 4 * c2p * 2);
        feat = std::max(feat, hs / 4 * (ws / 4) * cp * 2);
        f32 = std::max(f32, std::max(hs / 4 * (ws / 4) * cp * 4, f32_native_elems((int)(hs / 4), (int)(ws / 4), 2) * (cp / 64) * 4));
        t96 = std::max(t96, hs / 4 * (ws / 4) * 96 * 4);
        tmp = std::max(tmp, hs * ws * 6 * 4);
    }
    p.xin = take(xin);
    p.u0 = take(u0);
    p.a0 = take(a0);
    p.u1 = take(u1);
    p.featA = take(feat);
    p.featB = take(feat);
    p.f32A = take(f32);
    p.f32B = take(f32);
    p.t96 = take(t96);
    p.tmp = take(tmp);
    p.total = o;
    return p;
}

void run_conv(const fw_ifnet* n, const Conv& cv, const void* x, int h, int w, void* out, float* out_f32, int act, const float* res,
              const float* beta, int post_act, hipStream_t st, bool trunk = false, bool narrow = false, void* out_lo = nullptr) {
    const bool merged = n->merge_groups && cv.wall.p;
    narrow = narrow && merged && cv.nwall.p;
    for (const auto& g : cv.groups) {
        if (merged && g.off > 0) break;   // group 0's launch carries all of them
        ConvParams p{};
        p.in = x;
        p.in_cstride = cv.cin_pad;
        p.in_pstride = 32;
        p.out_pstride = 32;
        p.cin_chunks = n->skip_pad_chunks ? cv.cin_chunks : cv.cin_pad / 32;
        p.H = h;
        p.W = w;
        p.wpk = g.w.p;
        p.bias = (const float*)g.b.p;
        p.out = out;
        p.out_cstride = cv.cout_pad;
        p.out_coff = g.off;
        p.out_f32 = out_f32;
        p.out_lo = out_lo;
        p.res1 = res;
        p.s1 = 1.f;
        p.s2 = 1.f;
        p.act = act;
        p.chan_scale = beta ? beta + g.off : nullptr;
        p.post_act = post_act;
        p.f32_cstride = cv.cout_pad;
        p.f32_coff = g.off;
        if (trunk && n->native_trunk && cv.cout_pad % 64 == 0) {
            // the fp32 trunk of the ResConv chain in the accumulator-native layout (one region per 64-channel group): a KiB per
            // wave-instruction instead of 16 pixels x 64 bytes; only these epilogues ever read it
            const size_t ge = f32_native_elems(h, w, narrow ? 1 : 2), gi = (size_t)g.off / 64;
            p.f32_native = 1;
            p.f32_gstride = (long)ge;
            if (!merged) {
                if (p.out_f32) p.out_f32 += gi * ge;
                if (p.res1) p.res1 += gi * ge;
            }
        }
        if (merged) {
            p.wpk = cv.wall.p;
            p.bias = (const float*)cv.ball.p;
            p.n_groups = (int)cv.groups.size();
            p.wpk_gstride = (long)cv.gstride;
            if (narrow) {
                p.wpk = cv.nwall.p;
                p.n_groups = cv.cout_pad / 32;
                p.wpk_gstride = (long)cv.ngstride;
            }
        }
        launch_conv3x3(n->dt, narrow ? 1 : g.ct, res ? EPI_RESIDUAL : EPI_STORE, p, st);
    }
}

// lrelu(conv'(x) + x) on a split trunk: x = hi + lo (two typed NHWC tensors of 64 channels), conv' = the ResConv with beta folded in
void run_resconv_split(const fw_ifnet* n, const Conv& cv, const void* x_hi, const void* x_lo, int h, int w, void* out_hi, void* out_lo,
                       hipStream_t st) {
    const auto& g = cv.groups.at(0);
    ConvParams p{};
    p.in = x_hi;
    p.in_cstride = 64;
    p.in_pstride = 32;
    p.out_pstride = 32;
    p.cin_chunks = 2;
    p.H = h;
    p.W = w;
    p.wpk = g.w.p;
    p.bias = (const float*)g.b.p;
    p.out = out_hi;
    p.out_lo = out_lo;
    p.out_cstride = 64;
    p.s1 = 1.f;
    p.s2 = 1.f;
    p.post_act = 1;
    p.in_id_scale = 1.f;                                   // + hi, from the centre tap of the tile in LDS
    p.n_id = 2;                                            // + lo, channels [0, 32) and [32, 64)
    const long lo_off = (const char*)x_lo - (const char*)x_hi;
    p.chunk_off[0] = lo_off;
    p.chunk_off[1] = lo_off + 64;
    p.id_scale[0] = p.id_scale[1] = 1.f;
    launch_conv3x3(n->dt, 2, EPI_RESIDUAL_SPLIT, p, st);
}

void forward(fw_ifnet* n, const uint8_t* d0, const uint8_t* d1, int H, int W, float timestep, uint8_t* d_out, float* d_rgb,
             hipStream_t st) {
    const int Hp = pad_to(H, 32), Wp = pad_to(W, 32);
    const Plan pl = make_plan(H, W);
    char* ws = (char*)n->ws.p;
    float *I0 = (float*)(ws + pl.I0), *I1 = (float*)(ws + pl.I1), *flow = (float*)(ws + pl.flow), *mask = (float*)(ws + pl.mask);
    float *X = (float*)(ws + pl.X), *xin = (float*)(ws + pl.xin), *t96 = (float*)(ws + pl.t96), *tmp = (float*)(ws + pl.tmp);
    void *u0 = ws + pl.u0, *a0 = ws + pl.a0, *u1 = ws + pl.u1;
    launch_ifnet_u8_to_rgb(d0, H, W, Hp, Wp, I0, st);
    launch_ifnet_u8_to_rgb(d1, H, W, Hp, Wp, I1, st);
    for (int i = 0; i < NBLK; ++i) {
        const Block& b = n->blk[i];
        const bool first = i == 0;
        const int s = SC[i], hs = Hp / s, wsz = Wp / s;
        if (n->fuse_glue) {
            launch_ifnet_stage_input(n->dt, I0, I1, first ? nullptr : flow, first ? nullptr : mask, Hp, Wp, timestep, s, u0, b.conv00.cin_pad, st);
        } else {
            launch_ifnet_build_x(I0, I1, first ? nullptr : flow, first ? nullptr : mask, Hp, Wp, timestep, X, st);
            launch_resize_bilinear(X, Hp, Wp, first ? 7 : 8, xin, hs, wsz, b.cin, 0, 1.0f / s, 1.0f, st);
            if (!first) launch_resize_bilinear(flow, Hp, Wp, 4, xin, hs, wsz, b.cin, 8, 1.0f / s, 1.0f / s, st);
            launch_unshuffle2_cast(n->dt, xin, true, hs, wsz, b.cin, b.cin, u0, b.conv00.cin_pad, st);
        }
        run_conv(n, b.conv00, u0, hs / 2, wsz / 2, a0, nullptr, 1, nullptr, nullptr, 0, st);
        launch_unshuffle2_cast(n->dt, a0, false, hs / 2, wsz / 2, b.c2p, b.c2p, u1, b.conv01.cin_pad, st);
        const int hf = hs / 4, wf = wsz / 4;
        void *feat = ws + pl.featA, *nxt = ws + pl.featB;
        float *feat32 = (float*)(ws + pl.f32A), *nxt32 = (float*)(ws + pl.f32B);
        // few tiles (the low-resolution blocks): 32-channel groups - twice the workgroups, half the MFMAs each.  One choice for the whole
        // chain: the native fp32 trunk's layout follows the group width.
        const long tiles = (long)(f32_native_elems(hf, wf, 1) / (512 * 32));
        const bool narrow = n->narrow_groups && tiles * (b.cp / 64) < n->narrow_below;
        if (n->split_trunk && b.cp == 64 && !b.res_split[0].groups.empty()) {
            // the trunk as hi + lo (the fp32 trunk's buffers hold the lo tensors: they are twice as large)
            run_conv(n, b.conv01, u1, hf, wf, feat, nullptr, 1, nullptr, nullptr, 0, st, false, false, feat32);
            for (int j = 0; j < NRES; ++j) {
                run_resconv_split(n, b.res_split[j], feat, feat32, hf, wf, nxt, nxt32, st);
                std::swap(feat, nxt);
                std::swap(feat32, nxt32);
            }
        } else {
        run_conv(n, b.conv01, u1, hf, wf, feat, feat32, 1, nullptr, nullptr, 0, st, true, narrow);
        for (int j = 0; j < NRES; ++j) {   // ResConv: lrelu(conv(x) * beta + x)
            run_conv(n, b.res[j], feat, hf, wf, nxt, nxt32, 0, feat32, (const float*)b.beta[j].p, 1, st, true, narrow);
            std::swap(feat, nxt);
            std::swap(feat32, nxt32);
        }
        }
        run_conv(n, b.last, feat, hf, wf, nullptr, t96, 0, nullptr, nullptr, 0, st);
        if (n->fuse_glue) {
            launch_ifnet_accumulate_d2s(t96, hf, wf, 96, Hp, Wp, (float)s, flow, mask, first ? 1 : 0, st);
        } else {
            launch_depth_to_space4(t96, hf, wf, 96, tmp, st);
            launch_ifnet_accumulate(tmp, hs, wsz, Hp, Wp, (float)s, flow, mask, first ? 1 : 0, st);
        }
    }
    launch_ifnet_blend(I0, I1, flow, mask, Hp, Wp, H, W, d_out, d_rgb, st);
    FW_HIP_CHECK(hipGetLastError());
}

void drop_graphs(fw_ifnet* n) {
    for (auto& g : n->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    n->graphs.clear();
}

void need(size_t got, size_t want, const std::string& key) {
    if (got != want)
        throw Error(FW_ERR_INVALID, "fw_ifnet_set_tensor: '" + key + "' has " + std::to_string(got) + " elements, expected " + std::to_string(want));
}

}  // namespace

extern "C" {

int fw_ifnet_create(int device_id, int dtype, fw_ifnet** out) {
    if (!out) return fail(FW_ERR_INVALID, "fw_ifnet_create: out is NULL");
    *out = nullptr;
    if (dtype != FW_DTYPE_BF16 && dtype != FW_DTYPE_F16) return fail(FW_ERR_INVALID, "fw_ifnet_create: bad dtype");
    return guarded([&] {
        int nd = 0;
        FW_HIP_CHECK(hipGetDeviceCount(&nd));
        if (device_id < 0 || device_id >= nd) throw Error(FW_ERR_INVALID, "fw_ifnet_create: no such device");
        auto n = std::make_unique<fw_ifnet>();
        n->device = device_id;
        n->dt = (DType)dtype;
        if (const char* e = getenv("FW_IFNET_GRAPH")) n->graph_mode = atoi(e);
        if (const char* e = getenv("FW_IFNET_MERGE_GROUPS")) n->merge_groups = atoi(e) != 0;
        if (const char* e = getenv("FW_IFNET_FUSE_GLUE")) n->fuse_glue = atoi(e) != 0;
        if (const char* e = getenv("FW_IFNET_NATIVE_TRUNK")) n->native_trunk = atoi(e) != 0;
        if (const char* e = getenv("FW_IFNET_SPLIT_TRUNK")) n->split_trunk = atoi(e) != 0;
        if (const char* e = getenv("FW_IFNET_SKIP_PAD")) n->skip_pad_chunks = atoi(e) != 0;
        if (const char* e = getenv("FW_IFNET_NARROW")) n->narrow_groups = atoi(e) != 0;
        if (const char* e = getenv("FW_IFNET_NARROW_BELOW")) n->narrow_below = atol(e);
        for (int i = 0; i < NBLK; ++i) {
            Block& b = n->blk[i];
            b.c = CH[i];
            b.cin = i == 0 ? 7 : 12;
            b.c2p = pad_to(CH[i] / 2, 32);
            b.cp = pad_to(CH[i], 64);
        }
        *out = n.release();
    });
}

int fw_ifnet_set_tensor(fw_ifnet* n, const char* key_c, const float* data, size_t numel) {
    if (!n || !key_c || !data) return fail(FW_ERR_INVALID, "fw_ifnet_set_tensor: NULL argument");
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        const std::string key(key_c);
        int i = -1, used = 0;
        if (sscanf(key_c, "block%d.%n", &i, &used) != 1 || used <= 0 || i < 0 || i >= NBLK)
            throw Error(FW_ERR_INVALID, "fw_ifnet_set_tensor: unknown tensor '" + key + "'");
        Block& b = n->blk[i];
        const std::string rest = key.substr(used);
        const size_t c = b.c, cin = b.cin;
        auto set = [&](std::vector<float>& dst, size_t want, unsigned bit) {
            need(numel, want, key);
            dst.assign(data, data + numel);
            b.have |= 1u << bit;
            n->built = false;
        };
        if (rest == "conv0.0.0.weight") return set(b.h_w00, c / 2 * cin * 9, 0);
        if (rest == "conv0.0.0.bias") return set(b.h_b00, c / 2, 1);
        if (rest == "conv0.1.0.weight") return set(b.h_w01, c * (c / 2) * 9, 2);
        if (rest == "conv0.1.0.bias") return set(b.h_b01, c, 3);
        if (rest == "lastconv.0.weight") return set(b.h_wl, c * 24 * 16, 4);
        if (rest == "lastconv.0.bias") return set(b.h_bl, 24, 5);
        int j = -1, u2 = 0;
        if (sscanf(rest.c_str(), "convblock.%d.%n", &j, &u2) == 1 && u2 > 0 && j >= 0 && j < NRES) {
            const std::string r2 = rest.substr(u2);
            if (r2 == "conv.weight") return set(b.h_wr[j], c * c * 9, 6 + 3 * j);
            if (r2 == "conv.bias") return set(b.h_br[j], c, 7 + 3 * j);
            if (r2 == "beta") return set(b.h_beta[j], c, 8 + 3 * j);
        }
        throw Error(FW_ERR_INVALID, "fw_ifnet_set_tensor: unknown tensor '" + key + "'");
    });
}

int fw_ifnet_finalize(fw_ifnet* n) {
    if (!n) return fail(FW_ERR_INVALID, "fw_ifnet_finalize: NULL");
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        if (n->built) return;
        for (int i = 0; i < NBLK; ++i)
            if (n->blk[i].have != BLOCK_ALL)
                throw Error(FW_ERR_INVALID, "fw_ifnet_finalize: block" + std::to_string(i) + " is missing tensors");
        DevGuard dg(n->device);
        drop_graphs(n);
        for (int i = 0; i < NBLK; ++i) {
            Block& b = n->blk[i];
            const int c = b.c, cin = b.cin;
            b.conv00.build(n->dt, stride2_as_unshuffled(b.h_w00.data(), c / 2, cin), b.h_b00, c / 2, 4 * cin, pad_to(4 * cin, 32), b.c2p);
            // conv0.1 reads the c2p-padded output of conv0.0: widen its input channels to c2p before the stride-2 transform
            std::vector<float> w1((size_t)c * b.c2p * 9, 0.f);
            for (int co = 0; co < c; ++co)
                for (int ci = 0; ci < c / 2; ++ci)
                    for (int t = 0; t < 9; ++t) w1[((size_t)co * b.c2p + ci) * 9 + t] = b.h_w01[((size_t)co * (c / 2) + ci) * 9 + t];
            b.conv01.build(n->dt, stride2_as_unshuffled(w1.data(), c, b.c2p), b.h_b01, c, 4 * b.c2p, 4 * b.c2p, b.cp, true);
            b.conv01.cin_chunks = (4 * (c / 2) + 31) / 32;   // channel ci * 4 + sub of the unshuffled tensor: those of the padded ci >= c / 2 are zero
            for (int j = 0; j < NRES; ++j) {
                b.res[j].build(n->dt, b.h_wr[j], b.h_br[j], c, c, b.cp, b.cp, true);
                std::vector<float> beta(b.cp, 0.f);
                for (int k = 0; k < c; ++k) beta[k] = b.h_beta[j][k];
                upload(b.beta[j], beta.data(), beta.size() * 4);
                if (b.cp == 64) {
                    std::vector<float> ws(b.h_wr[j]), bs(b.h_br[j]);
                    for (int co = 0; co < c; ++co) {
                        for (size_t k = 0; k < (size_t)c * 9; ++k) ws[(size_t)co * c * 9 + k] *= b.h_beta[j][co];
                        bs[co] *= b.h_beta[j][co];
                    }
                    b.res_split[j].build(n->dt, ws, bs, c, c, b.cp, b.cp);
                }
            }
            std::vector<float> w3, b3;
            convtranspose_as_3x3(b.h_wl.data(), b.h_bl.data(), c, 24, &w3, &b3);
            if (n->fuse_glue) {
                // the accumulate kernel reads lastconv's output in place: channel (sub-position of the 4 x 4 block) * 6 + c6 instead of
                // c6 * 16 + sub-position, so that the 5 values a tap needs are contiguous (as they were in the depth-to-space copy)
                std::vector<float> wq(w3.size()), bq(b3.size());
                const size_t row = (size_t)c * 9;
                for (int nn = 0; nn < 96; ++nn) {
                    const int c6 = nn / 16, pos = nn % 16, np = pos * 6 + c6;
                    std::copy(w3.begin() + nn * row, w3.begin() + (nn + 1) * row, wq.begin() + np * row);
                    bq[np] = b3[nn];
                }
                w3.swap(wq);
                b3.swap(bq);
            }
            b.last.build(n->dt, w3, b3, 96, c, b.cp, 96);
        }
        n->built = true;
    });
}

size_t fw_ifnet_workspace_bytes(const fw_ifnet* n, int H, int W) {
    if (!n || H < 1 || W < 1) return 0;
    return make_plan(H, W).total;
}

/* 2 * MACs of one forward on the padded frame (the stride-2 convs counted as the 3x3 convs on unshuffled tensors that run) */
double fw_ifnet_flops(const fw_ifnet* n, int H, int W) {
    if (!n || H < 1 || W < 1) return 0.0;
    const double Hp = pad_to(H, 32), Wp = pad_to(W, 32);
    double mac = 0;
    for (int i = 0; i < NBLK; ++i) {
        const double hs = Hp / SC[i], ws = Wp / SC[i], c = CH[i], cin = i == 0 ? 7 : 12;
        mac += hs / 2 * ws / 2 * 9 * (4 * cin) * (c / 2);
        mac += hs / 4 * ws / 4 * (9 * (4 * c / 2) * c + NRES * 9 * c * c + 9 * c * 96);
    }
    return 2.0 * mac;
}

int fw_ifnet_interp_u8(fw_ifnet* n, const uint8_t* frame0, const uint8_t* frame1, int in_loc, int H, int W, float timestep,
                       uint8_t* out_bgr, int out_loc, float* out_rgb_f32, void* stream) {
    if (!n || !frame0 || !frame1) return fail(FW_ERR_INVALID, "fw_ifnet_interp_u8: NULL argument");
    if (!out_bgr && !out_rgb_f32) return fail(FW_ERR_INVALID, "fw_ifnet_interp_u8: no output requested");
    if (H < 1 || W < 1 || H > 16384 || W > 16384) return fail(FW_ERR_INVALID, "fw_ifnet_interp_u8: bad frame size");
    if ((in_loc != FW_HOST && in_loc != FW_DEVICE) || (out_loc != FW_HOST && out_loc != FW_DEVICE))
        return fail(FW_ERR_INVALID, "fw_ifnet_interp_u8: bad buffer location");
    int rc = fw_ifnet_finalize(n);
    if (rc != FW_OK) return rc;
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        DevGuard dg(n->device);
        hipStream_t st = (hipStream_t)stream;
        StreamOrder::Scope in_order(n->order, st);
        const Plan pl = make_plan(H, W);
        if (n->ws.bytes < pl.total) {
            FW_HIP_CHECK(hipDeviceSynchronize());
            drop_graphs(n);
            n->ws.release();
            FW_HIP_CHECK(hipMalloc(&n->ws.p, pl.total));
            n->ws.bytes = pl.total;
        }
        char* ws = (char*)n->ws.p;
        const size_t bytes = (size_t)H * W * 3;
        const uint8_t *d0 = frame0, *d1 = frame1;
        if (in_loc == FW_HOST) {
            FW_HIP_CHECK(hipMemcpyAsync(ws + pl.in0, frame0, bytes, hipMemcpyHostToDevice, st));
            FW_HIP_CHECK(hipMemcpyAsync(ws + pl.in1, frame1, bytes, hipMemcpyHostToDevice, st));
            d0 = (const uint8_t*)(ws + pl.in0);
            d1 = (const uint8_t*)(ws + pl.in1);
        }
        uint8_t* d_out = out_bgr;
        if (out_bgr && out_loc == FW_HOST) d_out = (uint8_t*)(ws + pl.out_u8);
        const bool graphed = n->warmed && n->graph_mode == 1;
        n->warmed = true;
        if (!graphed) {
            forward(n, d0, d1, H, W, timestep, d_out, out_rgb_f32, st);
        } else {
            fw_ifnet::GraphEntry* hit = nullptr;
            for (auto& g : n->graphs)
                if (g.H == H && g.W == W && g.t == timestep && g.a == d0 && g.b == d1 && g.out == d_out && g.rgb == out_rgb_f32) hit = &g;
            if (!hit) {
                if (n->graphs.size() >= 16) drop_graphs(n);
                (void)conv_zero_page();   // its first use allocates: not inside a capture
                hipStream_t cs = nullptr;
                FW_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
                fw_ifnet::GraphEntry e{H, W, timestep, d0, d1, d_out, out_rgb_f32, nullptr, nullptr};
                hipError_t err = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
                if (err == hipSuccess) {
                    try {
                        forward(n, d0, d1, H, W, timestep, d_out, out_rgb_f32, cs);
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
            FW_HIP_CHECK(hipMemcpyAsync(out_bgr, d_out, bytes, hipMemcpyDeviceToHost, st));
            FW_HIP_CHECK(hipStreamSynchronize(st));
        }
    });
}

int fw_ifnet_destroy(fw_ifnet* n) {
    if (!n) return FW_OK;
    { std::lock_guard<std::mutex> lk(n->mu); }   // a call in flight on another thread finishes first
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(n->device);
    (void)hipDeviceSynchronize();
    drop_graphs(n);
    for (auto& b : n->blk) {
        b.conv00.release();
        b.conv01.release();
        b.last.release();
        for (int j = 0; j < NRES; ++j) {
            b.res[j].release();
            b.res_split[j].release();
            b.beta[j].release();
        }
    }
    n->ws.release();
    if (prev >= 0) (void)hipSetDevice(prev);
    n->order.destroy();
    delete n;
    return FW_OK;
}

}  // extern "C"
