// Restormer - the reference's DEFAULT TAP model - as ONE engine behind the C-ABI: weights, workspace arena, launch sequencing.
//
// Reference call sites: src/framewright/processors/tap_denoise.py:299-333 (`Restormer(inp_channels=3, out_channels=3, dim=48,
// num_blocks=[4,6,6,8], num_refinement_blocks=4, heads=[1,2,4,8], ffn_expansion_factor=2.66, bias=False,
// LayerNorm_type='WithBias')`, weights under `params` / `state_dict`), :458 (the forward), :373-415 (pre / post-processing).
// The class lives in a third-party package that is absent here: the architecture follows SURVEY.md section A.4 and the oracle
// is oracle/restormer_ref.py (parity vs upstream unpinned).
//
// Round 1 sequenced the ~800 launches of a forward from Python over the building blocks (fw_layernorm_nhwc, fw_pointwise_nhwc,
// fw_dwconv3x3_nhwc, fw_attn_*, ...).  The same launches are issued here from C++: one mutex per handle, one workspace arena
// sized by a dry run of the sequencing (no allocator traffic per call), so a non-Python binder of include/framewright_hip.h gets
// the operator (fw_load_* / fw_denoise_u8 per model).
//
// Layout (unchanged): fp32 NHWC residual stream [pixels][pad64(c)]; LayerNorm writes operand-typed tensors for the GEMMs; q / k /
// v in one typed buffer at channel offsets 0 / cp / 2cp; the GDFN halves x1 / x2 at 0 / hp; the 1x1 and depthwise weights are
// re-laid on the host to match.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "fw_internal.h"
#include "../../include/framewright_hip.h"

using namespace fw;

namespace {

int pad_to(int n, int m) { return (n + m - 1) / m * m; }
// channels per pixel of the fp32 stream (and of every typed operand) at c real channels: whole 32-channel contraction chunks - 48 -> 64,
// 96 / 192 / 384 as they are (round 2 padded to 64: a third more bytes per pixel at 96 channels, where 20 of the 44 blocks run)
int stream_pad(int c) { return pad_to(c, 32); }

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

// [cout][k] fp32 -> packed pointwise fragments, cout padded to 32 and k to k_pad; returns the number of 32-channel tiles
int upload_pointwise(DType dt, DevBuf& b, const float* w, int cout, int k, int k_pad) {
    const int cp = pad_to(cout, 32);
    std::vector<float> wp((size_t)cp * k_pad, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int i = 0; i < k; ++i) wp[(size_t)co * k_pad + i] = w[(size_t)co * k + i];
    std::vector<uint16_t> pk(fw_pack_pointwise(dt, nullptr, cp, k_pad, nullptr));
    if (fw_pack_pointwise(dt, wp.data(), cp, k_pad, pk.data()) != pk.size()) throw Error(FW_ERR_INTERNAL, "fw_pack_pointwise failed");
    upload(b, pk.data(), pk.size() * 2);
    return cp / 32;
}

// a bias-free 3x3 convolution on chunk-planar typed input, split into launches of 64 output channels, fp32 NHWC out
struct Conv3 {
    int cin_pad = 0, cout_pad = 0;
    std::vector<DevBuf> groups;
    DevBuf wall;            // all groups in one buffer, gstride bytes apart: they run side by side in one launch (ConvParams::n_groups)
    size_t gstride = 0;
    void release() {
        for (auto& g : groups) g.release();
        groups.clear();
        wall.release();
        gstride = 0;
    }
    void build(DType dt, const float* w, int cout, int cin, int cin_pad_or_0) {
        release();
        cin_pad = cin_pad_or_0 ? cin_pad_or_0 : stream_pad(cin);
        cout_pad = pad_to(cout, 64);
        std::vector<float> wp((size_t)cout_pad * cin_pad * 9, 0.f);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < 9; ++t) wp[((size_t)co * cin_pad + ci) * 9 + t] = w[((size_t)co * cin + ci) * 9 + t];
        const int chunks = cin_pad / 32;
        std::vector<uint16_t> all;
        for (int off = 0; off < cout_pad; off += 64) {
            std::vector<uint16_t> pk(pack_conv3x3_weights(dt, nullptr, 64, cin_pad, 2, chunks, nullptr));
            pack_conv3x3_weights(dt, wp.data() + (size_t)off * cin_pad * 9, 64, cin_pad, 2, chunks, pk.data());
            groups.emplace_back();
            upload(groups.back(), pk.data(), pk.size() * 2);
            gstride = pk.size() * 2;
            all.insert(all.end(), pk.begin(), pk.end());
        }
        if (groups.size() > 1) upload(wall, all.data(), all.size() * 2);
    }
};

struct RBlock {
    int c = 0, cp = 0, heads = 0, ch = 0, hid = 0, hp = 0;
    DevBuf n1w, n1b, n2w, n2b, temp, qkv, qkv_dw, proj, pin, ffn_dw, pout;
    DevBuf proj_f32;               // project_out fp32 [c][c]: folded into the attention matrix per forward (fw_attn_proj_pack)
    DevBuf front_qkv, front_ffn;   // pw_dw_fused.hip parameter blocks (c = 48 / 96): norm1 + qkv + qkv_dwconv, norm2 + project_in + dwconv
    DevBuf qkv16, pin16;           // qkv / project_in in pack_pointwise_weights16's layout, rows padded to 256 (the staged levels: pointwise_gemm.hip)
    int qkv16_n = 0, pin16_n = 0;  // padded output channels
    int qkv_t = 0, proj_t = 0, pin_t = 0, pout_t = 0;
    void release() {
        for (DevBuf* b : {&n1w, &n1b, &n2w, &n2b, &temp, &qkv, &qkv_dw, &proj, &pin, &ffn_dw, &pout, &front_qkv, &front_ffn, &proj_f32, &qkv16, &pin16}) b->release();
    }
};

struct Stage {
    std::string name;
    int n, c, heads;
};

// bump allocator over the engine's workspace; `plan` = dry run of the sequencing that only measures the peak
struct Arena {
    char* base = nullptr;
    size_t top = 0, peak = 0;
    bool plan = false;
    void* take(size_t bytes) {
        const size_t at = top;
        top += (bytes + 255) / 256 * 256;
        if (top > peak) peak = top;
        return base + at;   // plan mode: base == nullptr, the pointer is never dereferenced or launched on
    }
};

}  // namespace

struct fw_restormer {
    int device = 0;
    fw::StreamOrder order;   // device-side ordering of forwards enqueued on different streams (fw_internal.h)
    DType dt = DT_F16;
    int dim = 48, nblk[4] = {4, 6, 6, 8}, nref = 4, heads[4] = {1, 2, 4, 8};
    double ffn = 2.66;
    std::mutex mu;
    std::vector<Stage> stages;
    std::map<std::string, size_t> want;             // tensor key -> element count
    std::map<std::string, std::vector<float>> host; // as set, until finalize()
    std::map<std::string, RBlock> blocks;           // "encoder_level1.0." -> device weights
    std::map<std::string, Conv3> convs;
    DevBuf red3, red2, conv_bias, ones;
    int red3_t = 0, red2_t = 0;
    bool built = false;
    bool merge_groups = true; // the 64-channel output groups of a 3x3 conv in one launch (FW_REST_MERGE_GROUPS=0: A/B)
    bool gemm16 = true;       // qkv / project_in of the 192- / 384-channel levels on the pipelined GEMM kernel (FW_REST_GEMM16=0: A/B)
    bool qk_direct = true;    // the fused qkv front writes q / k in the Gram kernel's operand layout (FW_REST_QK_DIRECT=0: pixel-major + transpose pass)
    bool merge_proj = true;   // project_out folded into the attention matrix: one GEMM pass instead of two (FW_REST_MERGE_PROJ=0: A/B)
    bool fuse_front = true;   // LayerNorm + 1x1 + depthwise 3x3 (+ GDFN gate) of the 48- / 96-channel blocks as one kernel (FW_REST_FUSE_FRONT=0: A/B)
    DevBuf ws;
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

// a building block failed: its message is already in fw_last_error()
void chk(int status) {
    if (status != FW_OK) throw Error(status, fw::last_error_ref());
}

const char* GLOBAL_CONV3[] = {"patch_embed.proj.weight", "down1_2.body.0.weight", "down2_3.body.0.weight", "down3_4.body.0.weight",
                              "up4_3.body.0.weight", "up3_2.body.0.weight", "up2_1.body.0.weight", "output.weight"};

void forward(fw_restormer* n, Arena& A, const uint8_t* d_in, int H, int W, uint8_t* d_out, float* d_rgb, hipStream_t st_) {
    void* st = (void*)st_;
    const int dt = (int)n->dt;
    const bool run = !A.plan;
    auto f32 = [&](size_t elems) { return (float*)A.take(elems * 4); };
    auto typ = [&](size_t elems) { return A.take(elems * 2); };
    auto zero = [&](void* p, size_t bytes) {
        if (run) FW_HIP_CHECK(hipMemsetAsync(p, 0, bytes, st_));
    };
#define RUN(expr)            \
    do {                     \
        if (run) chk(expr);  \
    } while (0)

    // 3x3 conv of the fp32 stream x [h*w][cin_pad] -> fp32 y [h*w][cout_pad] (y allocated by the caller)
    auto conv3 = [&](const Conv3& cv, const float* x, int h, int w, float* y) {
        const long M = (long)h * w;
        const size_t mark = A.top;
        void* xp = typ((size_t)cv.cin_pad * M);
        RUN(fw_f32_to_planar(dt, x, M, cv.cin_pad, xp, st));
        if (cv.wall.p && n->merge_groups && cv.groups.size() <= 64) {
            // the 64-channel output groups side by side in one launch (the up convs have 3 / 6 / 12 of them on 64 ... 16 k pixels)
            ConvParams p{};
            p.in = xp; p.in_cstride = 32; p.in_pstride = M * 32; p.out_pstride = 32; p.cin_chunks = cv.cin_pad / 32; p.H = h; p.W = w;
            p.wpk = cv.wall.p; p.bias = (const float*)n->conv_bias.p; p.out_f32 = y; p.s1 = p.s2 = 1.f; p.f32_cstride = cv.cout_pad;
            p.n_groups = (int)cv.groups.size(); p.wpk_gstride = (long)cv.gstride;
            if (run) launch_conv3x3(n->dt, 2, EPI_STORE, p, st_);
        } else {
            for (size_t g = 0; g < cv.groups.size(); ++g)
                RUN(fw_conv3x3_nhwc_ex(dt, xp, 32, M * 32, cv.cin_pad / 32, h, w, cv.groups[g].p, (const float*)n->conv_bias.p, 2, 0, 0, nullptr,
                                       1.f, nullptr, 1.f, nullptr, 0, cv.cout_pad, (int)g * 64, nullptr, 32, 0, 0, y, st));
        }
        A.top = mark;
    };

    // one transformer block on the fp32 stream x [h*w][cp], updated in place
    auto block = [&](const RBlock& b, float* x, int h, int w) {
        const long M = (long)h * w;
        const int c = b.c, cp = b.cp, hp = b.hp, heads = b.heads, ch = b.ch;
        const size_t mark = A.top;
        void* t = typ((size_t)M * cp);
        const int nq = pad_to(3 * cp, 64);          // output channels of the fused qkv kernel: whole 64-channel chunks (q | k | v | zeros)
        void* qkv2 = typ((size_t)M * nq);
        const void* v = (const char*)qkv2 + (size_t)2 * cp * 2;   // v as the attention GEMM reads it: [M][v_ld] typed
        long v_ld = 3 * cp;
        // norm -> 1x1 -> depthwise 3x3 of the 48- / 96-channel blocks: one kernel, the 3c / 5.3c-channel tensor stays in LDS
        auto front = [&](const DevBuf& blocks, int n_out, int mode, void* out, long ldo) {
            PwDwParams f{};
            f.x = x; f.ldx = cp; f.H = h; f.W = w; f.cin = c; f.ln_eps = 1e-5f; f.blocks = blocks.p; f.n_chunks = n_out / 64; f.mode = mode;
            f.out = out; f.ldo = ldo;
            if (run) launch_pw_dw(n->dt, f, st_);
        };
        float* attn = f32((size_t)heads * ch * ch);
        float* aws = f32(fw_attn_workspace_floats(heads, ch));
        if (b.front_qkv.p) {
            // q and k leave the kernel already in the Gram kernel's operand layout (no pixel-major copy of them, no transpose pass)
            // (q | k: 2 cp channel rows per pixel group = whole chunks at 64 and at 96 channels; v: the chunks behind them, from channel 0 of qkv2)
            const long Mp = pw_dw_transposed_pixels(h, w);
            void* qT = n->qk_direct ? typ((size_t)Mp * 2 * cp) : nullptr;
            PwDwParams f{};
            f.x = x; f.ldx = cp; f.H = h; f.W = w; f.cin = c; f.ln_eps = 1e-5f; f.blocks = b.front_qkv.p; f.n_chunks = nq / 64; f.mode = PWDW_NONE;
            f.out = qkv2; f.ldo = n->qk_direct ? nq - 2 * cp : nq; f.qT = qT; f.t_chunks = 2 * cp / 64; f.t_ld = 2 * cp;
            if (run) launch_pw_dw(n->dt, f, st_);
            if (n->qk_direct) {
                v = qkv2;
                v_ld = nq - 2 * cp;
                if (run)
                    launch_attn_matrix_from_transposed(n->dt, qT, (const char*)qT + (size_t)cp * 8 * 2, Mp, 2 * cp, heads, ch, (const float*)b.temp.p, aws, attn,
                                                       st_);
            } else {
                v_ld = nq;
                void* scratch = typ(fw_attn_qk_scratch_elems(M, heads, ch));
                RUN(fw_attn_matrix_mfma(dt, qkv2, nq, M, cp, heads, ch, (const float*)b.temp.p, aws, scratch, attn, st));
            }
        } else {
            RUN(fw_layernorm_nhwc(dt, x, cp, M, c, (const float*)b.n1w.p, (const float*)b.n1b.p, 1e-5f, t, cp, cp, st));
            // a typed-store 1x1 GEMM of the staged levels on the pipelined kernel (256-channel tiles: the output rows are qld wide)
            auto gemm = [&](const DevBuf& w, const DevBuf& w16, int tiles, int n16, void* out, long ld) {
                PointwiseParams p{};
                p.a = t; p.lda = cp; p.M = M; p.K = cp; p.wpk = w.p; p.wpk16 = w16.p; p.N_tiles = w16.p ? n16 / 32 : tiles; p.mode = PW_STORE;
                p.out_typed = out; p.ldo = ld;
                if (run) launch_pointwise(n->dt, p, st_);
            };
            const long qld = b.qkv16.p ? b.qkv16_n : 3 * cp;
            void* qkv = typ((size_t)M * qld);
            gemm(b.qkv, b.qkv16, b.qkv_t, b.qkv16_n, qkv, qld);
            RUN(fw_dwconv3x3_nhwc(dt, qkv, qld, h, w, 3 * cp, (const float*)b.qkv_dw.p, 0, qkv2, 3 * cp, st));
            void* scratch = typ(fw_attn_qk_scratch_elems(M, heads, ch));
            RUN(fw_attn_matrix_mfma(dt, qkv2, 3 * cp, M, cp, heads, ch, (const float*)b.temp.p, aws, scratch, attn, st));
        }
        // attn @ v as a 1x1 convolution with the block-diagonal attention matrix on the MFMA GEMM
        void* apk = typ(fw_pack_pointwise(dt, nullptr, cp, cp, nullptr));
        if (n->merge_proj) {
            // x += (project_out . attn) v: the two matrices are multiplied first (c x c x ch MACs), the pixels see one GEMM
            RUN(fw_attn_proj_pack(dt, attn, (const float*)b.proj_f32.p, heads, ch, cp, b.proj_t, apk, st));
            RUN(fw_pointwise_nhwc(dt, v, 0, v_ld, M, cp, apk, nullptr, b.proj_t, nullptr, 0, x, cp, x, (const float*)n->ones.p, st));
        } else {
            RUN(fw_attn_pack(dt, attn, heads, ch, cp, apk, st));
            RUN(fw_pointwise_nhwc(dt, v, 0, v_ld, M, cp, apk, nullptr, cp / 32, t, cp, nullptr, 0, nullptr, nullptr, st));
            RUN(fw_pointwise_nhwc(dt, t, 0, cp, M, cp, b.proj.p, nullptr, b.proj_t, nullptr, 0, x, cp, x, (const float*)n->ones.p, st));
        }
        void* g2 = typ((size_t)M * hp);
        if (b.front_ffn.p) {
            front(b.front_ffn, 2 * hp, PWDW_GATE_GELU, g2, hp);
        } else {
            RUN(fw_layernorm_nhwc(dt, x, cp, M, c, (const float*)b.n2w.p, (const float*)b.n2b.p, 1e-5f, t, cp, cp, st));
            const long gld = b.pin16.p ? b.pin16_n : 2 * hp;
            void* g = typ((size_t)M * gld);
            {
                PointwiseParams p{};
                p.a = t; p.lda = cp; p.M = M; p.K = cp; p.wpk = b.pin.p; p.wpk16 = b.pin16.p; p.N_tiles = b.pin16.p ? b.pin16_n / 32 : b.pin_t; p.mode = PW_STORE;
                p.out_typed = g; p.ldo = gld;
                if (run) launch_pointwise(n->dt, p, st_);
            }
            RUN(fw_dwconv3x3_nhwc(dt, g, gld, h, w, 2 * hp, (const float*)b.ffn_dw.p, 1, g2, hp, st));
        }
        RUN(fw_pointwise_nhwc(dt, g2, 0, hp, M, hp, b.pout.p, nullptr, b.pout_t, nullptr, 0, x, cp, x, (const float*)n->ones.p, st));
        A.top = mark;
    };
    auto stage = [&](int si, float* x, int h, int w) {
        const Stage& s = n->stages[si];
        for (int i = 0; i < s.n; ++i) block(n->blocks.at(s.name + "." + std::to_string(i) + "."), x, h, w);
    };
    // conv3x3 (c -> c/2) + PixelUnshuffle(2): fp32 [h*w][stream_pad(c)] -> fp32 [(h/2)*(w/2)][stream_pad(2c)]
    auto down = [&](const char* key, const float* x, int h, int w, int c) {
        const Conv3& cv = n->convs.at(key);
        const size_t ostride = stream_pad(2 * c);
        float* o = f32((size_t)(h / 2) * (w / 2) * ostride);
        zero(o, (size_t)(h / 2) * (w / 2) * ostride * 4);
        const size_t mark = A.top;
        float* y = f32((size_t)h * w * cv.cout_pad);
        conv3(cv, x, h, w, y);
        RUN(fw_pixel_shuffle2_f32(y, cv.cout_pad, h / 2, w / 2, c / 2, o, (long)ostride, 0, 1, st));
        A.top = mark;
        return o;
    };
    // cat([PixelShuffle(2)(conv3x3 (c -> 2c)(x)), skip]): fp32 [(2h)*(2w)][stream_pad(c)], c/2 + c/2 channels
    auto up_cat = [&](const char* key, const float* x, int h, int w, int c, const float* skip, int skip_stride) {
        const Conv3& cv = n->convs.at(key);
        const size_t ostride = stream_pad(c);
        float* o = f32((size_t)4 * h * w * ostride);
        zero(o, (size_t)4 * h * w * ostride * 4);
        const size_t mark = A.top;
        float* y = f32((size_t)h * w * cv.cout_pad);
        conv3(cv, x, h, w, y);
        RUN(fw_pixel_shuffle2_f32(y, cv.cout_pad, h, w, c / 2, o, (long)ostride, 0, 0, st));
        A.top = mark;
        RUN(fw_copy_channels_f32(skip, skip_stride, (long)4 * h * w, c / 2, o, (long)ostride, c / 2, st));
        return o;
    };
    auto reduce = [&](const DevBuf& wpk, int tiles, const float* x, int x_stride, long M, int cin) {
        float* o = f32((size_t)M * 32 * tiles);
        RUN(fw_pointwise_nhwc(dt, x, 1, x_stride, M, cin, wpk.p, nullptr, tiles, nullptr, 0, o, 32 * tiles, nullptr, nullptr, st));
        return o;
    };

    const int d = n->dim;
    void* x0 = typ((size_t)H * W * 32);
    RUN(fw_u8_to_nhwc(dt, d_in, H, W, x0, 32, st));
    const Conv3& pe = n->convs.at("patch_embed.proj.weight");
    float* e1 = f32((size_t)H * W * pe.cout_pad);
    for (size_t g = 0; g < pe.groups.size(); ++g)
        RUN(fw_conv3x3_nhwc_ex(dt, x0, 32, 0, 1, H, W, pe.groups[g].p, (const float*)n->conv_bias.p, 2, 0, 0, nullptr, 1.f, nullptr, 1.f, nullptr, 0,
                               pe.cout_pad, (int)g * 64, nullptr, 32, 0, 0, e1, st));
    stage(0, e1, H, W);
    float* e2 = down("down1_2.body.0.weight", e1, H, W, d);
    stage(1, e2, H / 2, W / 2);
    float* e3 = down("down2_3.body.0.weight", e2, H / 2, W / 2, 2 * d);
    stage(2, e3, H / 4, W / 4);
    float* lat = down("down3_4.body.0.weight", e3, H / 4, W / 4, 4 * d);
    stage(3, lat, H / 8, W / 8);
    float* c3 = up_cat("up4_3.body.0.weight", lat, H / 8, W / 8, 8 * d, e3, stream_pad(4 * d));
    float* d3 = reduce(n->red3, n->red3_t, c3, stream_pad(8 * d), (long)(H / 4) * (W / 4), 8 * d);
    stage(4, d3, H / 4, W / 4);
    float* c2 = up_cat("up3_2.body.0.weight", d3, H / 4, W / 4, 4 * d, e2, stream_pad(2 * d));
    float* d2 = reduce(n->red2, n->red2_t, c2, stream_pad(4 * d), (long)(H / 2) * (W / 2), 4 * d);
    stage(5, d2, H / 2, W / 2);
    float* d1 = up_cat("up2_1.body.0.weight", d2, H / 2, W / 2, 2 * d, e1, stream_pad(d));
    stage(6, d1, H, W);
    stage(7, d1, H, W);
    const Conv3& oc = n->convs.at("output.weight");
    float* y = f32((size_t)H * W * oc.cout_pad);
    conv3(oc, d1, H, W, y);
    RUN(fw_tap_post_u8(d_in, y, H, W, W, oc.cout_pad, d_out, d_rgb, st));
#undef RUN
}

}  // namespace

extern "C" {

int fw_restormer_create(int device_id, int dim, const int* num_blocks, int num_refinement_blocks, const int* heads,
                        double ffn_expansion_factor, int dtype, fw_restormer** out) {
    if (!out || !num_blocks || !heads) return fail(FW_ERR_INVALID, "fw_restormer_create: NULL argument");
    *out = nullptr;
    if (dim != 48) return fail(FW_ERR_INVALID, "fw_restormer_create: dim must be 48 (per-head widths 48 / 96 are what the attention kernels take)");
    if (dtype != FW_DTYPE_BF16 && dtype != FW_DTYPE_F16) return fail(FW_ERR_INVALID, "fw_restormer_create: bad dtype");
    if (num_refinement_blocks < 0 || num_refinement_blocks > 64 || !(ffn_expansion_factor >= 1.0 && ffn_expansion_factor <= 8.0))
        return fail(FW_ERR_INVALID, "fw_restormer_create: bad refinement count / expansion factor");
    return guarded([&] {
        int nd = 0;
        FW_HIP_CHECK(hipGetDeviceCount(&nd));
        if (device_id < 0 || device_id >= nd) throw Error(FW_ERR_INVALID, "fw_restormer_create: no such device");
        auto n = std::make_unique<fw_restormer>();
        n->device = device_id;
        n->dt = (DType)dtype;
        n->dim = dim;
        n->nref = num_refinement_blocks;
        n->ffn = ffn_expansion_factor;
        if (const char* e = getenv("FW_REST_FUSE_FRONT")) n->fuse_front = atoi(e) != 0;
        if (const char* e = getenv("FW_REST_MERGE_PROJ")) n->merge_proj = atoi(e) != 0;
        if (const char* e = getenv("FW_REST_QK_DIRECT")) n->qk_direct = atoi(e) != 0;
        if (const char* e = getenv("FW_REST_MERGE_GROUPS")) n->merge_groups = atoi(e) != 0;
        if (const char* e = getenv("FW_REST_GEMM16")) n->gemm16 = atoi(e) != 0;
        for (int i = 0; i < 4; ++i) {
            if (num_blocks[i] < 0 || num_blocks[i] > 64 || heads[i] < 1) throw Error(FW_ERR_INVALID, "fw_restormer_create: bad block / head counts");
            n->nblk[i] = num_blocks[i];
            n->heads[i] = heads[i];
        }
        const int d = dim;
        n->stages = {{"encoder_level1", n->nblk[0], d, n->heads[0]},     {"encoder_level2", n->nblk[1], 2 * d, n->heads[1]},
                     {"encoder_level3", n->nblk[2], 4 * d, n->heads[2]}, {"latent", n->nblk[3], 8 * d, n->heads[3]},
                     {"decoder_level3", n->nblk[2], 4 * d, n->heads[2]}, {"decoder_level2", n->nblk[1], 2 * d, n->heads[1]},
                     {"decoder_level1", n->nblk[0], 2 * d, n->heads[0]}, {"refinement", n->nref, 2 * d, n->heads[0]}};
        for (const auto& s : n->stages) {
            if (s.c % s.heads || (s.c / s.heads != 48 && s.c / s.heads != 96))
                throw Error(FW_ERR_INVALID, "fw_restormer_create: channels per head must be 48 or 96");
            const size_t c = s.c, hid = (size_t)(int)(s.c * n->ffn);
            for (int i = 0; i < s.n; ++i) {
                const std::string p = s.name + "." + std::to_string(i) + ".";
                n->want[p + "norm1.body.weight"] = c;
                n->want[p + "norm1.body.bias"] = c;
                n->want[p + "attn.temperature"] = s.heads;
                n->want[p + "attn.qkv.weight"] = 3 * c * c;
                n->want[p + "attn.qkv_dwconv.weight"] = 3 * c * 9;
                n->want[p + "attn.project_out.weight"] = c * c;
                n->want[p + "norm2.body.weight"] = c;
                n->want[p + "norm2.body.bias"] = c;
                n->want[p + "ffn.project_in.weight"] = 2 * hid * c;
                n->want[p + "ffn.dwconv.weight"] = 2 * hid * 9;
                n->want[p + "ffn.project_out.weight"] = c * hid;
            }
        }
        const size_t D = d;
        n->want["patch_embed.proj.weight"] = D * 3 * 9;
        n->want["down1_2.body.0.weight"] = D / 2 * D * 9;
        n->want["down2_3.body.0.weight"] = D * 2 * D * 9;
        n->want["down3_4.body.0.weight"] = 2 * D * 4 * D * 9;
        n->want["up4_3.body.0.weight"] = 16 * D * 8 * D * 9;
        n->want["reduce_chan_level3.weight"] = 4 * D * 8 * D;
        n->want["up3_2.body.0.weight"] = 8 * D * 4 * D * 9;
        n->want["reduce_chan_level2.weight"] = 2 * D * 4 * D;
        n->want["up2_1.body.0.weight"] = 4 * D * 2 * D * 9;
        n->want["output.weight"] = 3 * 2 * D * 9;
        *out = n.release();
    });
}

int fw_restormer_set_tensor(fw_restormer* n, const char* key_c, const float* data, size_t numel) {
    if (!n || !key_c || !data) return fail(FW_ERR_INVALID, "fw_restormer_set_tensor: NULL argument");
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        const std::string key(key_c);
        auto it = n->want.find(key);
        if (it == n->want.end()) throw Error(FW_ERR_INVALID, "fw_restormer_set_tensor: unknown tensor '" + key + "'");
        if (numel != it->second)
            throw Error(FW_ERR_INVALID, "fw_restormer_set_tensor: '" + key + "' has " + std::to_string(numel) + " elements, expected " +
                                            std::to_string(it->second));
        n->host[key].assign(data, data + numel);
        n->built = false;
    });
}

int fw_restormer_finalize(fw_restormer* n) {
    if (!n) return fail(FW_ERR_INVALID, "fw_restormer_finalize: NULL");
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        if (n->built) return;
        for (const auto& kv : n->want)
            if (!n->host.count(kv.first)) throw Error(FW_ERR_INVALID, "fw_restormer_finalize: missing " + kv.first);
        DevGuard dg(n->device);
        const DType dt = n->dt;
        for (const auto& s : n->stages) {
            const int c = s.c, cp = stream_pad(c), hid = (int)(c * n->ffn), hp = pad_to(hid, 32);
            for (int i = 0; i < s.n; ++i) {
                const std::string p = s.name + "." + std::to_string(i) + ".";
                RBlock& b = n->blocks[p];
                b.c = c;
                b.cp = cp;
                b.heads = s.heads;
                b.ch = c / s.heads;
                b.hid = hid;
                b.hp = hp;
                auto H = [&](const char* k) -> const std::vector<float>& { return n->host.at(p + k); };
                upload(b.n1w, H("norm1.body.weight").data(), (size_t)c * 4);
                upload(b.n1b, H("norm1.body.bias").data(), (size_t)c * 4);
                upload(b.n2w, H("norm2.body.weight").data(), (size_t)c * 4);
                upload(b.n2b, H("norm2.body.bias").data(), (size_t)c * 4);
                upload(b.temp, H("attn.temperature").data(), (size_t)s.heads * 4);
                // qkv: rows re-laid to q @ 0, k @ cp, v @ 2cp (1x1 and depthwise alike)
                const auto& wq = H("attn.qkv.weight");
                const auto& dws = H("attn.qkv_dwconv.weight");
                std::vector<float> wqkv((size_t)3 * cp * c, 0.f), wdw((size_t)3 * cp * 9, 0.f);
                for (int t = 0; t < 3; ++t)
                    for (int r = 0; r < c; ++r) {
                        std::copy(wq.begin() + (size_t)(t * c + r) * c, wq.begin() + (size_t)(t * c + r + 1) * c, wqkv.begin() + (size_t)(t * cp + r) * c);
                        std::copy(dws.begin() + (size_t)(t * c + r) * 9, dws.begin() + (size_t)(t * c + r + 1) * 9, wdw.begin() + (size_t)(t * cp + r) * 9);
                    }
                b.qkv_t = upload_pointwise(dt, b.qkv, wqkv.data(), 3 * cp, c, cp);
                upload(b.qkv_dw, wdw.data(), wdw.size() * 4);
                const bool fused = n->fuse_front && pw_dw_eligible(c, PWDW_NONE);
                auto upload_front = [&](DevBuf& dst, const std::vector<float>& w1, const char* lnw, const char* lnb, const std::vector<float>& dw, int N,
                                        int gate) {
                    std::vector<char> pk(pack_pw_dw_blocks(dt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, c, gate, nullptr));
                    pack_pw_dw_blocks(dt, w1.data(), nullptr, H(lnw).data(), H(lnb).data(), dw.data(), nullptr, N, c, gate, pk.data());
                    upload(dst, pk.data(), pk.size());
                };
                if (fused) {
                    const int nq = pad_to(3 * cp, 64);   // whole 64-channel chunks: zero rows behind v at 96 channels (288 -> 320)
                    std::vector<float> wq_f(wqkv), wdw_f(wdw);
                    wq_f.resize((size_t)nq * c, 0.f);
                    wdw_f.resize((size_t)nq * 9, 0.f);
                    upload_front(b.front_qkv, wq_f, "norm1.body.weight", "norm1.body.bias", wdw_f, nq, 0);
                }
                auto upload16 = [&](DevBuf& dst, const std::vector<float>& w, int rows, int* n_out) {   // [rows][c] -> [pad256(rows)][cp], 16-layout
                    const int np = pad_to(rows, 256);
                    std::vector<float> wp((size_t)np * cp, 0.f);
                    for (int r = 0; r < rows; ++r) std::copy(w.begin() + (size_t)r * c, w.begin() + (size_t)(r + 1) * c, wp.begin() + (size_t)r * cp);
                    std::vector<uint16_t> pk(pack_pointwise_weights16(dt, nullptr, np, cp, 0, nullptr));
                    pack_pointwise_weights16(dt, wp.data(), np, cp, 0, pk.data());
                    upload(dst, pk.data(), pk.size() * 2);
                    *n_out = np;
                };
                const bool gemm16 = n->gemm16 && !fused && cp % 64 == 0 && cp >= 128;
                if (gemm16) upload16(b.qkv16, wqkv, 3 * cp, &b.qkv16_n);
                b.proj_t = upload_pointwise(dt, b.proj, H("attn.project_out.weight").data(), c, c, cp);
                upload(b.proj_f32, H("attn.project_out.weight").data(), (size_t)c * c * 4);
                // GDFN: x1 rows @ 0, x2 rows @ hp
                const auto& wi = H("ffn.project_in.weight");
                const auto& di = H("ffn.dwconv.weight");
                std::vector<float> wi2((size_t)2 * hp * c, 0.f), di2((size_t)2 * hp * 9, 0.f);
                for (int half = 0; half < 2; ++half)
                    for (int r = 0; r < hid; ++r) {
                        std::copy(wi.begin() + (size_t)(half * hid + r) * c, wi.begin() + (size_t)(half * hid + r + 1) * c, wi2.begin() + (size_t)(half * hp + r) * c);
                        std::copy(di.begin() + (size_t)(half * hid + r) * 9, di.begin() + (size_t)(half * hid + r + 1) * 9, di2.begin() + (size_t)(half * hp + r) * 9);
                    }
                b.pin_t = upload_pointwise(dt, b.pin, wi2.data(), 2 * hp, c, cp);
                if (gemm16) upload16(b.pin16, wi2, 2 * hp, &b.pin16_n);
                upload(b.ffn_dw, di2.data(), di2.size() * 4);
                if (fused && hp % 32 == 0 && (2 * hp) % 64 == 0) upload_front(b.front_ffn, wi2, "norm2.body.weight", "norm2.body.bias", di2, 2 * hp, 1);
                b.pout_t = upload_pointwise(dt, b.pout, H("ffn.project_out.weight").data(), c, hid, hp);
            }
        }
        const int d = n->dim;
        const int cins[] = {3, d, 2 * d, 4 * d, 8 * d, 4 * d, 2 * d, 2 * d};
        const int couts[] = {d, d / 2, d, 2 * d, 16 * d, 8 * d, 4 * d, 3};
        for (int k = 0; k < 8; ++k)
            n->convs[GLOBAL_CONV3[k]].build(dt, n->host.at(GLOBAL_CONV3[k]).data(), couts[k], cins[k], k == 0 ? 32 : 0);
        n->red3_t = upload_pointwise(dt, n->red3, n->host.at("reduce_chan_level3.weight").data(), 4 * d, 8 * d, 8 * d);
        n->red2_t = upload_pointwise(dt, n->red2, n->host.at("reduce_chan_level2.weight").data(), 2 * d, 4 * d, 4 * d);
        std::vector<float> z(64 * 64, 0.f), o(2048, 1.f);   // zero bias for up to 64 output-channel groups in one launch
        upload(n->conv_bias, z.data(), z.size() * 4);
        upload(n->ones, o.data(), o.size() * 4);
        n->host.clear();
        n->built = true;
    });
}

size_t fw_restormer_workspace_bytes(fw_restormer* n, int H, int W) {
    if (!n || H < 8 || W < 8 || (H & 7) || (W & 7) || !n->built) return 0;
    Arena A;
    A.plan = true;
    try {
        forward(n, A, nullptr, H, W, nullptr, nullptr, nullptr);
    } catch (...) {
        return 0;
    }
    return A.peak + (size_t)H * W * 3 * 2 + 512;
}

int fw_restormer_denoise_u8(fw_restormer* n, const uint8_t* in_bgr, int in_loc, int H, int W, uint8_t* out_bgr, int out_loc,
                            float* out_rgb_f32, void* stream) {
    if (!n || !in_bgr) return fail(FW_ERR_INVALID, "fw_restormer_denoise_u8: NULL argument");
    if (!out_bgr && !out_rgb_f32) return fail(FW_ERR_INVALID, "fw_restormer_denoise_u8: no output requested");
    if (H < 8 || W < 8 || H > 16384 || W > 16384 || (H & 7) || (W & 7))
        return fail(FW_ERR_INVALID, "fw_restormer_denoise_u8: frame sizes must be multiples of 8 (the network's three PixelUnshuffles)");
    if ((in_loc != FW_HOST && in_loc != FW_DEVICE) || (out_loc != FW_HOST && out_loc != FW_DEVICE))
        return fail(FW_ERR_INVALID, "fw_restormer_denoise_u8: bad buffer location");
    int rc = fw_restormer_finalize(n);
    if (rc != FW_OK) return rc;
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        DevGuard dg(n->device);
        hipStream_t st = (hipStream_t)stream;
        StreamOrder::Scope in_order(n->order, st);
        const size_t frame = ((size_t)H * W * 3 + 255) / 256 * 256;
        Arena P;
        P.plan = true;
        forward(n, P, nullptr, H, W, nullptr, nullptr, nullptr);
        const size_t total = 2 * frame + P.peak;
        if (n->ws.bytes < total) {
            FW_HIP_CHECK(hipDeviceSynchronize());
            n->ws.release();
            FW_HIP_CHECK(hipMalloc(&n->ws.p, total));
            n->ws.bytes = total;
        }
        char* base = (char*)n->ws.p;
        const uint8_t* d_in = in_bgr;
        if (in_loc == FW_HOST) {
            FW_HIP_CHECK(hipMemcpyAsync(base, in_bgr, (size_t)H * W * 3, hipMemcpyHostToDevice, st));
            d_in = (const uint8_t*)base;
        }
        uint8_t* d_out = out_bgr;
        if (out_bgr && out_loc == FW_HOST) d_out = (uint8_t*)(base + frame);
        Arena A;
        A.base = base + 2 * frame;
        forward(n, A, d_in, H, W, d_out, out_rgb_f32, st);
        if (out_bgr && out_loc == FW_HOST) {
            FW_HIP_CHECK(hipMemcpyAsync(out_bgr, d_out, (size_t)H * W * 3, hipMemcpyDeviceToHost, st));
            FW_HIP_CHECK(hipStreamSynchronize(st));
        }
    });
}

int fw_restormer_destroy(fw_restormer* n) {
    if (!n) return FW_OK;
    { std::lock_guard<std::mutex> lk(n->mu); }   // a call in flight on another thread finishes first
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(n->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : n->blocks) kv.second.release();
    for (auto& kv : n->convs) kv.second.release();
    for (DevBuf* b : {&n->red3, &n->red2, &n->conv_bias, &n->ones, &n->ws}) b->release();
    if (prev >= 0) (void)hipSetDevice(prev);
    n->order.destroy();
    delete n;
    return FW_OK;
}

}  // extern "C"
