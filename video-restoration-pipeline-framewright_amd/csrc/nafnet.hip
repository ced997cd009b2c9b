// NAFNet forward for the TAP temporal-denoise path on one MI355X: weights, workspace, launch sequencing.
//
// Reference call sites: src/framewright/processors/tap_denoise.py:335-364 (`NAFNet(img_channel=3, width=64,
// middle_blk_num=12, enc_blk_nums=[2,2,4,8], dec_blk_nums=[2,2,2,2])`, weights under `params`/`state_dict`),
// :373-415 (pre/post-processing: BGR->RGB, /255, ... np.clip(x*255, 0, 255).astype(uint8) — truncation), :458 (the
// forward).  The network itself lives in a third-party package that is absent here; the architecture follows
// SURVEY.md §A.3 and the oracle is oracle/nafnet_ref.py ("parity vs upstream unpinned").
#include <cmath>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include "fw_internal.h"
#include "../../include/framewright_hip.h"

using namespace fw;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct Block {
    int c = 0;
    // device weights
    DevBuf n1w, n1b, n2w, n2b, beta, gamma;            // fp32 [c]
    DevBuf w1, b1, w3, b3, w4, b4, w5, b5;             // packed pointwise weights + fp32 biases
    DevBuf w1g, w3g, w4g, w5g;                         // the same in pack_pointwise_weights16's layout (c >= 256: the GEMM kernel)
    DevBuf wdw, bdw;                                   // depthwise fp32 [2c][9], [2c]
    DevBuf wsca, bsca;                                 // fp32 [c][c], [c]
    DevBuf front;                                      // pw_dw_fused.hip (c = 64 / 128): parameter blocks of norm1 + conv1 + conv2, built by
                                                       // fw_nafnet_finalize from the host copies below
    std::vector<float> h_w1, h_b1, h_n1w, h_n1b, h_wdw, h_bdw;
    DevBuf tail128;                                    // naf_tail128.hip (c = 128): the eight weight blocks of the block's second half
    std::vector<float> h_w3, h_b3, h_beta, h_n2w, h_n2b, h_w4, h_b4, h_w5, h_b5, h_gamma;
    unsigned have = 0;                                 // bit per tensor
};
constexpr unsigned BLOCK_ALL = (1u << 18) - 1;

struct Level {
    DevBuf w, b;  // down: packed [2c][4c] + bias[2c]; up: packed [2c_hi][c_hi] (no bias)
    bool have_w = false, have_b = false;
};

}  // namespace

struct fw_nafnet {
    int device = 0;
    fw::StreamOrder order;   // device-side ordering of forwards enqueued on different streams (fw_internal.h)
    DType dt = DT_BF16;
    int width = 64, middle = 12;
    int nlev = 4;
    int enc[8] = {0}, dec[8] = {0};
    std::mutex mu;
    std::vector<std::vector<Block>> encoders, decoders;
    std::vector<Block> middle_blks;
    std::vector<Level> downs, ups;
    DevBuf intro_w, intro_b, ending_w, ending_b;
    bool have_intro_w = false, have_intro_b = false, have_end_w = false, have_end_b = false;
    DevBuf ws;
    bool fuse_ln = true;   // LayerNorm2d inside the staging pass of the GEMM that follows it at width 64 (FW_NAF_FUSE_LN=0: A/B)
    bool gemm = true;      // 1x1 convs of the levels with >= 256 channels on pointwise_gemm.hip (FW_NAF_GEMM=0: A/B)
    bool fuse_tail = true; // conv3 .. conv5 of a width-64 block as one kernel (FW_NAF_FUSE_TAIL=0: A/B)
    bool fuse_front = true; // norm1 + conv1 + depthwise conv + gate of a 64- / 128-channel block as one kernel (FW_NAF_FUSE_FRONT=0)
    // hipGraph replay of a forward (BASELINE configs[4]: "hipGraph-captured per-frame stages"), keyed by everything a captured launch
    // sequence bakes in (FW_NAF_GRAPH=1; off by default: a caller that hands over fresh buffers every frame would re-capture every frame)
    int graph_mode = 0;
    bool warmed = false;
    struct GraphEntry {
        int H, W;
        const void* in;
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

void drop_graphs(fw_nafnet* n) {
    for (auto& g : n->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    n->graphs.clear();
}

void upload(DevBuf& b, const void* src, size_t bytes) {
    b.release();
    FW_HIP_CHECK(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    FW_HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
}

void upload_pointwise(DType dt, DevBuf& b, const float* w, int cout, int K) {
    std::vector<uint16_t> pk(pack_pointwise_weights(dt, nullptr, cout, K, nullptr));
    pack_pointwise_weights(dt, w, cout, K, pk.data());
    upload(b, pk.data(), pk.size() * 2);
}

// the deep levels run their 1x1 convs on the pipelined GEMM kernel (pointwise_gemm.hip): a second packing of the same weights
void upload_pointwise16(DType dt, DevBuf& b, const float* w, int cout, int K, int gate) {
    std::vector<uint16_t> pk(pack_pointwise_weights16(dt, nullptr, cout, K, gate, nullptr));
    pack_pointwise_weights16(dt, w, cout, K, gate, pk.data());
    upload(b, pk.data(), pk.size() * 2);
}

// norm1 + conv1 + conv2 of a 64- / 128-channel block as parameter blocks of pw_dw_fused.hip (norm1's affine part folded into conv1)
void build_front(fw_nafnet* n, Block& bl) {
    const int c = bl.c;
    std::vector<char> pk(pack_pw_dw_blocks(n->dt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 2 * c, c, 1, nullptr));
    pack_pw_dw_blocks(n->dt, bl.h_w1.data(), bl.h_b1.data(), bl.h_n1w.data(), bl.h_n1b.data(), bl.h_wdw.data(), bl.h_bdw.data(), 2 * c, c, 1, pk.data());
    upload(bl.front, pk.data(), pk.size());
}

// conv3 .. conv5 of a 128-channel block as the weight blocks of naf_tail128.hip (norm2's affine part folded into conv4)
void build_tail128(fw_nafnet* n, Block& bl) {
    std::vector<char> pk(naf_tail128_block_bytes());
    pack_naf_tail128_blocks(n->dt, bl.h_w3.data(), bl.h_b3.data(), bl.h_beta.data(), bl.h_n2w.data(), bl.h_n2b.data(), bl.h_w4.data(), bl.h_b4.data(),
                            bl.h_w5.data(), bl.h_b5.data(), bl.h_gamma.data(), pk.data());
    upload(bl.tail128, pk.data(), pk.size());
}

void upload_conv3(DType dt, DevBuf& b, const float* w, int cout, int cin) {
    const int ct = (cout + 31) / 32, ch = (cin + 31) / 32;
    std::vector<uint16_t> pk(pack_conv3x3_weights(dt, nullptr, cout, cin, ct, ch, nullptr));
    pack_conv3x3_weights(dt, w, cout, cin, ct, ch, pk.data());
    upload(b, pk.data(), pk.size() * 2);
}

void upload_padded_bias(DevBuf& b, const float* src, int n, int padded) {
    std::vector<float> v(padded, 0.f);
    for (int i = 0; i < n; ++i) v[i] = src[i];
    upload(b, v.data(), v.size() * 4);
}

// which block does "encoders.1.0." / "middle_blks.3." / "decoders.2.1." name?
Block* find_block(fw_nafnet* n, const std::string& key, std::string* rest) {
    int a = -1, b = -1, used = 0;
    if (sscanf(key.c_str(), "encoders.%d.%d.%n", &a, &b, &used) == 2 && used > 0) {
        if (a < 0 || a >= n->nlev || b < 0 || b >= (int)n->encoders[a].size()) return nullptr;
        *rest = key.substr(used);
        return &n->encoders[a][b];
    }
    if (sscanf(key.c_str(), "decoders.%d.%d.%n", &a, &b, &used) == 2 && used > 0) {
        if (a < 0 || a >= n->nlev || b < 0 || b >= (int)n->decoders[a].size()) return nullptr;
        *rest = key.substr(used);
        return &n->decoders[a][b];
    }
    if (sscanf(key.c_str(), "middle_blks.%d.%n", &a, &used) == 1 && used > 0) {
        if (a < 0 || a >= (int)n->middle_blks.size()) return nullptr;
        *rest = key.substr(used);
        return &n->middle_blks[a];
    }
    return nullptr;
}

void need(size_t got, size_t want, const std::string& key) {
    if (got != want)
        throw Error(FW_ERR_INVALID, "fw_nafnet_set_tensor: '" + key + "' has " + std::to_string(got) + " elements, expected " +
                                        std::to_string(want));
}

void set_block_tensor(fw_nafnet* n, Block& bl, const std::string& name, const std::string& key, const float* d,
                      size_t numel) {
    const int c = bl.c;
    auto mark = [&](int bit) { bl.have |= 1u << bit; };
    const bool front = n->fuse_front && pw_dw_eligible(c, PWDW_GATE_MUL);
    const bool tail = n->fuse_tail && c == 128;
    if (name == "norm1.weight") { need(numel, c, key); upload(bl.n1w, d, c * 4); if (front) { bl.h_n1w.assign(d, d + numel); bl.front.release(); } mark(0); }
    else if (name == "norm1.bias") { need(numel, c, key); upload(bl.n1b, d, c * 4); if (front) { bl.h_n1b.assign(d, d + numel); bl.front.release(); } mark(1); }
    else if (name == "norm2.weight") { need(numel, c, key); upload(bl.n2w, d, c * 4); if (tail) { bl.h_n2w.assign(d, d + numel); bl.tail128.release(); } mark(2); }
    else if (name == "norm2.bias") { need(numel, c, key); upload(bl.n2b, d, c * 4); if (tail) { bl.h_n2b.assign(d, d + numel); bl.tail128.release(); } mark(3); }
    else if (name == "beta") { need(numel, c, key); upload(bl.beta, d, c * 4); if (tail) { bl.h_beta.assign(d, d + numel); bl.tail128.release(); } mark(4); }
    else if (name == "gamma") { need(numel, c, key); upload(bl.gamma, d, c * 4); if (tail) { bl.h_gamma.assign(d, d + numel); bl.tail128.release(); } mark(5); }
    else if (name == "conv1.weight") { need(numel, (size_t)2 * c * c, key); upload_pointwise(n->dt, bl.w1, d, 2 * c, c); if (n->gemm && c >= 256) upload_pointwise16(n->dt, bl.w1g, d, 2 * c, c, 0); if (front) { bl.h_w1.assign(d, d + numel); bl.front.release(); } mark(6); }
    else if (name == "conv1.bias") { need(numel, 2 * c, key); upload(bl.b1, d, 2 * c * 4); if (front) { bl.h_b1.assign(d, d + numel); bl.front.release(); } mark(7); }
    else if (name == "conv2.weight") { need(numel, (size_t)2 * c * 9, key); upload(bl.wdw, d, (size_t)2 * c * 9 * 4); if (front) { bl.h_wdw.assign(d, d + numel); bl.front.release(); } mark(8); }
    else if (name == "conv2.bias") { need(numel, 2 * c, key); upload(bl.bdw, d, 2 * c * 4); if (front) { bl.h_bdw.assign(d, d + numel); bl.front.release(); } mark(9); }
    else if (name == "conv3.weight") { need(numel, (size_t)c * c, key); upload_pointwise(n->dt, bl.w3, d, c, c); if (n->gemm && c >= 256) upload_pointwise16(n->dt, bl.w3g, d, c, c, 0); if (tail) { bl.h_w3.assign(d, d + numel); bl.tail128.release(); } mark(10); }
    else if (name == "conv3.bias") { need(numel, c, key); upload(bl.b3, d, c * 4); if (tail) { bl.h_b3.assign(d, d + numel); bl.tail128.release(); } mark(11); }
    else if (name == "conv4.weight") { need(numel, (size_t)2 * c * c, key); upload_pointwise(n->dt, bl.w4, d, 2 * c, c); if (n->gemm && c >= 256) upload_pointwise16(n->dt, bl.w4g, d, 2 * c, c, 1); if (tail) { bl.h_w4.assign(d, d + numel); bl.tail128.release(); } mark(12); }
    else if (name == "conv4.bias") { need(numel, 2 * c, key); upload(bl.b4, d, 2 * c * 4); if (tail) { bl.h_b4.assign(d, d + numel); bl.tail128.release(); } mark(13); }
    else if (name == "conv5.weight") { need(numel, (size_t)c * c, key); upload_pointwise(n->dt, bl.w5, d, c, c); if (n->gemm && c >= 256) upload_pointwise16(n->dt, bl.w5g, d, c, c, 0); if (tail) { bl.h_w5.assign(d, d + numel); bl.tail128.release(); } mark(14); }
    else if (name == "conv5.bias") { need(numel, c, key); upload(bl.b5, d, c * 4); if (tail) { bl.h_b5.assign(d, d + numel); bl.tail128.release(); } mark(15); }
    else if (name == "sca.1.weight") { need(numel, (size_t)c * c, key); upload(bl.wsca, d, (size_t)c * c * 4); mark(16); }
    else if (name == "sca.1.bias") { need(numel, c, key); upload(bl.bsca, d, c * 4); mark(17); }
    else throw Error(FW_ERR_INVALID, "fw_nafnet_set_tensor: unknown tensor '" + key + "'");
}

struct Plan {
    int Hp, Wp;
    size_t in_u8, out_u8, img32, S[8], T1, T2, T3, csum, sca, w3s, cat64, rgb, total;
};

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

Plan make_plan(const fw_nafnet* n, int H, int W) {
    Plan p{};
    const int mult = 1 << n->nlev;
    p.Hp = (H + mult - 1) / mult * mult;
    p.Wp = (W + mult - 1) / mult * mult;
    const size_t M0 = (size_t)p.Hp * p.Wp;
    size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += up256(b); return at; };
    p.in_u8 = take((size_t)H * W * 3);
    p.out_u8 = take((size_t)H * W * 3);
    p.img32 = take(M0 * 32 * 2);
    for (int l = 0; l <= n->nlev; ++l) p.S[l] = take((M0 >> (2 * l)) * ((size_t)n->width << l) * 4);
    p.T1 = take(M0 * n->width * 2);
    p.T2 = take(M0 * n->width * 2 * 2);
    p.T3 = take(M0 * n->width * 2);
    p.csum = take((size_t)1024 * 1024 * 4);  // dwconv partial sums [<=1024 blocks][<=1024 channels]
    p.sca = take(2 * 1024 * 4);  // SCA scale [C] + pooled mean [C]
    p.w3s = take((size_t)1024 * 1024 * 2);  // conv3's packed weights scaled by the SCA factors (<= 1024 x 1024)
    p.cat64 = take(M0 * n->width * 2);
    p.rgb = take(M0 * 3 * 4);
    p.total = o;
    return p;
}

void run_block(fw_nafnet* n, const Block& b, float* S, int H, int W, char* ws, const Plan& pl, hipStream_t st) {
    const int c = b.c;
    const long M = (long)H * W;
    void* T1 = ws + pl.T1;
    void* T2 = ws + pl.T2;
    void* T3 = ws + pl.T3;
    float* csum = (float*)(ws + pl.csum);
    float* sca = (float*)(ws + pl.sca);
    // x = conv1(norm1(inp)); at width 64 (the full-resolution level: four fifths of the LayerNorm bytes) the LayerNorm runs
    // inside the GEMM's staging pass
    const bool fuse_ln = c == 64 && n->fuse_ln;
    PointwiseParams p{};
    const bool front = b.front.p != nullptr;
    if (front) {
        // norm1, conv1, the depthwise conv and the gate in one kernel: the 2c-channel tensor stays in LDS (pw_dw_fused.hip)
        PwDwParams f{};
        f.x = S; f.ldx = c; f.H = H; f.W = W; f.cin = c; f.ln_eps = 1e-6f; f.blocks = b.front.p; f.n_chunks = 2 * c / 64; f.mode = PWDW_GATE_MUL;
        f.out = T3; f.ldo = c; f.partial = csum;
        launch_pw_dw(n->dt, f, st);
        launch_sca(csum, pw_dw_blocks(H, W), M, c, (const float*)b.wsca.p, (const float*)b.bsca.p, sca, st);
    } else {
    if (fuse_ln) {
        p.a = S; p.a_f32 = 1; p.ln_w = (const float*)b.n1w.p; p.ln_b = (const float*)b.n1b.p; p.ln_eps = 1e-6f;
    } else {
        launch_layernorm2d(n->dt, S, M, c, (const float*)b.n1w.p, (const float*)b.n1b.p, T1, st);
        p.a = T1;
    }
    p.lda = c; p.M = M; p.K = c; p.wpk = b.w1.p; p.wpk16 = b.w1g.p; p.bias = (const float*)b.b1.p; p.N_tiles = 2 * c / 32;
    p.mode = PW_STORE; p.out_typed = T2; p.ldo = 2 * c;
    launch_pointwise(n->dt, p, st);
    // x = SimpleGate(conv2(x)); pooled sums for SCA
    launch_dwconv3x3_gate(n->dt, T2, H, W, c, (const float*)b.wdw.p, (const float*)b.bdw.p, T3, csum, st);
    launch_sca(csum, dwconv_blocks(H, W, c), M, c, (const float*)b.wsca.p, (const float*)b.bsca.p, sca, st);
    }
    if (b.tail128.p) {
        // conv3 .. conv5 of a 128-channel block in one pass: its 128 KB of weights stream through LDS (naf_tail128.hip)
        NafTail128Params t{};
        t.x = T3; t.ldx = c; t.stream = S; t.lds_ = c; t.M = M; t.ln_eps = 1e-6f; t.blocks = b.tail128.p; t.w3_scratch = ws + pl.w3s;
        launch_naf_tail128(n->dt, t, sca, st);
        return;
    }
    if (c == 64 && n->fuse_tail) {
        // the rest of the block in one pass over the stream (nn_ops.hip naf_tail64_kernel)
        launch_naf_tail64(n->dt, T3, sca, S, M, b.w3.p, (const float*)b.b3.p, (const float*)b.beta.p, (const float*)b.n2w.p,
                          (const float*)b.n2b.p, 1e-6f, b.w4.p, (const float*)b.b4.p, b.w5.p, (const float*)b.b5.p,
                          (const float*)b.gamma.p, st);
        return;
    }
    // y = inp + conv3(x * sca) * beta
    p = PointwiseParams{};
    p.a = T3; p.lda = c; p.M = M; p.K = c; p.a_scale = sca; p.wpk = b.w3.p; p.bias = (const float*)b.b3.p; p.N_tiles = c / 32;
    p.mode = PW_RESIDUAL; p.out_f32 = S; p.res_f32 = S; p.ldf = c; p.chan_scale = (const float*)b.beta.p;
    if (b.w3g.p) {   // the GEMM kernel does not touch its activations: the SCA factors go into a scaled copy of the weights
        launch_pw16_scale_weights(n->dt, b.w3g.p, sca, c, c, ws + pl.w3s, st);
        p.wpk16 = ws + pl.w3s;
        p.a_scale = nullptr;
    }
    launch_pointwise(n->dt, p, st);
    // x = conv5(SimpleGate(conv4(norm2(y)))) ; out = y + x * gamma
    p = PointwiseParams{};
    if (fuse_ln) {
        p.a = S; p.a_f32 = 1; p.ln_w = (const float*)b.n2w.p; p.ln_b = (const float*)b.n2b.p; p.ln_eps = 1e-6f;
    } else {
        launch_layernorm2d(n->dt, S, M, c, (const float*)b.n2w.p, (const float*)b.n2b.p, T1, st);
        p.a = T1;
    }
    p.lda = c; p.M = M; p.K = c; p.wpk = b.w4.p; p.wpk16 = b.w4g.p; p.bias = (const float*)b.b4.p; p.N_tiles = 2 * c / 32;
    p.mode = PW_GATE; p.out_typed = T3; p.ldo = c;
    launch_pointwise(n->dt, p, st);
    p = PointwiseParams{};
    p.a = T3; p.lda = c; p.M = M; p.K = c; p.wpk = b.w5.p; p.wpk16 = b.w5g.p; p.bias = (const float*)b.b5.p; p.N_tiles = c / 32;
    p.mode = PW_RESIDUAL; p.out_f32 = S; p.res_f32 = S; p.ldf = c; p.chan_scale = (const float*)b.gamma.p;
    launch_pointwise(n->dt, p, st);
}

void forward(fw_nafnet* n, const uint8_t* d_in, int H, int W, uint8_t* d_out, float* d_rgb, hipStream_t st) {
    const Plan pl = make_plan(n, H, W);
    char* ws = (char*)n->ws.p;
    const int Hp = pl.Hp, Wp = pl.Wp;
    const long M0 = (long)Hp * Wp;
    float* S[8];
    for (int l = 0; l <= n->nlev; ++l) S[l] = (float*)(ws + pl.S[l]);

    // pre-process (tap_denoise.py:373-397) + zero pad to a multiple of 2^levels (NAFNet.check_image_size)
    launch_u8_to_nhwc_padded(n->dt, d_in, H, W, Hp, Wp, ws + pl.img32, st);
    // intro 3x3 (3 -> width), fp32 NHWC residual stream
    {
        ConvParams p{};
        p.in = ws + pl.img32; p.in_cstride = 32; p.in_pstride = 32; p.H = Hp; p.W = Wp; p.cin_chunks = 1;
        p.wpk = n->intro_w.p; p.bias = (const float*)n->intro_b.p; p.out_f32 = S[0]; p.s1 = p.s2 = 1.f;
        launch_conv3x3(n->dt, n->width / 32, EPI_STORE, p, st);
    }
    int h = Hp, w = Wp;
    for (int l = 0; l < n->nlev; ++l) {
        for (const Block& b : n->encoders[l]) run_block(n, b, S[l], h, w, ws, pl, st);
        // down: 2x2 stride 2, c -> 2c
        const int c = n->width << l;
        PointwiseParams p{};
        p.a = S[l]; p.a_f32 = 1; p.lda = c; p.M = (long)(h / 2) * (w / 2); p.K = 4 * c; p.gather2x2 = 1; p.Win = w; p.Cin = c;
        p.wpk = n->downs[l].w.p; p.bias = (const float*)n->downs[l].b.p; p.N_tiles = 2 * c / 32; p.mode = PW_STORE;
        p.out_f32 = S[l + 1]; p.ldf = 2 * c;
        launch_pointwise(n->dt, p, st);
        h /= 2;
        w /= 2;
    }
    for (const Block& b : n->middle_blks) run_block(n, b, S[n->nlev], h, w, ws, pl, st);
    for (int i = 0; i < n->nlev; ++i) {
        const int l = n->nlev - 1 - i;          // target level
        const int chi = n->width << (l + 1);    // channels of the coarser level
        // x = PixelShuffle(conv1x1(x)) + skip, written in place into the skip buffer
        PointwiseParams p{};
        p.a = S[l + 1]; p.a_f32 = 1; p.lda = chi; p.M = (long)h * w; p.K = chi; p.Win = w;
        p.wpk = n->ups[i].w.p; p.bias = nullptr; p.N_tiles = 2 * chi / 32; p.mode = PW_SHUFFLE_UP;
        p.out_f32 = S[l]; p.res_f32 = S[l]; p.ldf = chi / 2;
        launch_pointwise(n->dt, p, st);
        h *= 2;
        w *= 2;
        for (const Block& b : n->decoders[i]) run_block(n, b, S[l], h, w, ws, pl, st);
    }
    // ending 3x3 (width -> 3) + inp, crop, quantise (tap_denoise.py:399-415)
    launch_f32_to_planar(n->dt, S[0], M0, n->width, ws + pl.cat64, st);
    {
        ConvParams p{};
        p.in = ws + pl.cat64; p.in_cstride = 32; p.in_pstride = M0 * 32; p.H = Hp; p.W = Wp; p.cin_chunks = n->width / 32;
        p.wpk = n->ending_w.p; p.bias = (const float*)n->ending_b.p; p.out_rgb = (float*)(ws + pl.rgb); p.img_H = Hp;
        p.img_W = Wp; p.s1 = p.s2 = 1.f;
        launch_conv3x3(n->dt, 1, EPI_IMAGE, p, st);
    }
    launch_tap_post(d_in, (const float*)(ws + pl.rgb), H, W, Wp, 3, d_out, d_rgb, st);
}

}  // namespace

extern "C" {

int fw_nafnet_create(int device_id, int width, int middle_blk_num, const int* enc_blk_nums, const int* dec_blk_nums,
                     int num_levels, int dtype, fw_nafnet** out) {
    if (!out || !enc_blk_nums || !dec_blk_nums) return fail(FW_ERR_INVALID, "fw_nafnet_create: NULL argument");
    *out = nullptr;
    if ((width != 32 && width != 64) || (width << num_levels) > 1024)
        return fail(FW_ERR_INVALID, "fw_nafnet_create: width must be 32 or 64 with width*2^levels <= 1024");
    if (num_levels < 1 || num_levels > 6 || middle_blk_num < 0 || middle_blk_num > 64)
        return fail(FW_ERR_INVALID, "fw_nafnet_create: bad level/block counts");
    if (dtype != FW_DTYPE_BF16 && dtype != FW_DTYPE_F16) return fail(FW_ERR_INVALID, "fw_nafnet_create: bad dtype");
    return guarded([&] {
        int nd = 0;
        FW_HIP_CHECK(hipGetDeviceCount(&nd));
        if (device_id < 0 || device_id >= nd) throw Error(FW_ERR_INVALID, "fw_nafnet_create: no such device");
        auto n = std::make_unique<fw_nafnet>();
        if (const char* e = getenv("FW_NAF_FUSE_LN")) n->fuse_ln = atoi(e) != 0;
        if (const char* e = getenv("FW_NAF_GEMM")) n->gemm = atoi(e) != 0;
        if (const char* e = getenv("FW_NAF_FUSE_TAIL")) n->fuse_tail = atoi(e) != 0;
        if (const char* e = getenv("FW_NAF_FUSE_FRONT")) n->fuse_front = atoi(e) != 0;
        if (const char* e = getenv("FW_NAF_GRAPH")) n->graph_mode = atoi(e);
        n->device = device_id;
        n->dt = (DType)dtype;
        n->width = width;
        n->middle = middle_blk_num;
        n->nlev = num_levels;
        n->encoders.resize(num_levels);
        n->decoders.resize(num_levels);
        n->downs.resize(num_levels);
        n->ups.resize(num_levels);
        for (int l = 0; l < num_levels; ++l) {
            if (enc_blk_nums[l] < 0 || enc_blk_nums[l] > 64 || dec_blk_nums[l] < 0 || dec_blk_nums[l] > 64)
                throw Error(FW_ERR_INVALID, "fw_nafnet_create: bad block count");
            n->enc[l] = enc_blk_nums[l];
            n->dec[l] = dec_blk_nums[l];
            n->encoders[l].resize(enc_blk_nums[l]);
            for (auto& b : n->encoders[l]) b.c = width << l;
            // decoders[i] works at level nlev-1-i
            n->decoders[l].resize(dec_blk_nums[l]);
            for (auto& b : n->decoders[l]) b.c = width << (num_levels - 1 - l);
        }
        n->middle_blks.resize(middle_blk_num);
        for (auto& b : n->middle_blks) b.c = width << num_levels;
        *out = n.release();
    });
}

int fw_nafnet_set_tensor(fw_nafnet* n, const char* key_c, const float* data, size_t numel) {
    if (!n || !key_c || !data) return fail(FW_ERR_INVALID, "fw_nafnet_set_tensor: NULL argument");
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        DevGuard dg(n->device);
        if (!n->graphs.empty()) {   // captured forwards hold the addresses of the weights being replaced
            FW_HIP_CHECK(hipDeviceSynchronize());
            drop_graphs(n);
        }
        const std::string key(key_c);
        const int w = n->width;
        if (key == "intro.weight") { need(numel, (size_t)w * 3 * 9, key); upload_conv3(n->dt, n->intro_w, data, w, 3); n->have_intro_w = true; return; }
        if (key == "intro.bias") { need(numel, w, key); upload_padded_bias(n->intro_b, data, w, 64 > w ? 64 : w); n->have_intro_b = true; return; }
        if (key == "ending.weight") { need(numel, (size_t)3 * w * 9, key); upload_conv3(n->dt, n->ending_w, data, 3, w); n->have_end_w = true; return; }
        if (key == "ending.bias") { need(numel, 3, key); upload_padded_bias(n->ending_b, data, 3, 32); n->have_end_b = true; return; }
        int a = -1, used = 0;
        if (sscanf(key_c, "downs.%d.%n", &a, &used) == 1 && used > 0) {
            if (a < 0 || a >= n->nlev) throw Error(FW_ERR_INVALID, "fw_nafnet_set_tensor: bad level in '" + key + "'");
            const int c = w << a;
            const std::string rest = key.substr(used);
            if (rest == "weight") {
                need(numel, (size_t)2 * c * c * 4, key);
                // [2c][c][2][2] -> [2c][k = (dy*2+dx)*c + ci]
                std::vector<float> r((size_t)2 * c * 4 * c);
                for (int co = 0; co < 2 * c; ++co)
                    for (int ci = 0; ci < c; ++ci)
                        for (int s = 0; s < 4; ++s) r[(size_t)co * 4 * c + (size_t)s * c + ci] = data[((size_t)co * c + ci) * 4 + s];
                upload_pointwise(n->dt, n->downs[a].w, r.data(), 2 * c, 4 * c);
                n->downs[a].have_w = true;
            } else if (rest == "bias") {
                need(numel, 2 * c, key);
                upload(n->downs[a].b, data, (size_t)2 * c * 4);
                n->downs[a].have_b = true;
            } else throw Error(FW_ERR_INVALID, "fw_nafnet_set_tensor: unknown tensor '" + key + "'");
            return;
        }
        if (sscanf(key_c, "ups.%d.0.%n", &a, &used) == 1 && used > 0) {
            if (a < 0 || a >= n->nlev) throw Error(FW_ERR_INVALID, "fw_nafnet_set_tensor: bad level in '" + key + "'");
            if (key.substr(used) != "weight") throw Error(FW_ERR_INVALID, "fw_nafnet_set_tensor: unknown tensor '" + key + "'");
            const int chi = w << (n->nlev - a);  // ups[a] takes the channels of level nlev-a
            need(numel, (size_t)2 * chi * chi, key);
            // conv out channel co*4 + sub  ->  kernel order sub*(chi/2) + co     (PixelShuffle(2) un-interleaved)
            const int cup = chi / 2;
            std::vector<float> r((size_t)2 * chi * chi);
            for (int co = 0; co < cup; ++co)
                for (int s = 0; s < 4; ++s)
                    for (int k = 0; k < chi; ++k) r[((size_t)s * cup + co) * chi + k] = data[((size_t)co * 4 + s) * chi + k];
            upload_pointwise(n->dt, n->ups[a].w, r.data(), 2 * chi, chi);
            n->ups[a].have_w = true;
            return;
        }
        std::string rest;
        Block* bl = find_block(n, key, &rest);
        if (!bl) throw Error(FW_ERR_INVALID, "fw_nafnet_set_tensor: unknown tensor '" + key + "'");
        set_block_tensor(n, *bl, rest, key, data, numel);
    });
}

int fw_nafnet_finalize(fw_nafnet* n) {
    if (!n) return fail(FW_ERR_INVALID, "fw_nafnet_finalize: NULL");
    std::lock_guard<std::mutex> lk(n->mu);
    if (!n->have_intro_w || !n->have_intro_b || !n->have_end_w || !n->have_end_b)
        return fail(FW_ERR_INVALID, "fw_nafnet_finalize: intro/ending missing");
    for (int l = 0; l < n->nlev; ++l) {
        if (!n->downs[l].have_w || !n->downs[l].have_b) return fail(FW_ERR_INVALID, "fw_nafnet_finalize: missing downs." + std::to_string(l));
        if (!n->ups[l].have_w) return fail(FW_ERR_INVALID, "fw_nafnet_finalize: missing ups." + std::to_string(l));
        for (size_t j = 0; j < n->encoders[l].size(); ++j)
            if (n->encoders[l][j].have != BLOCK_ALL)
                return fail(FW_ERR_INVALID, "fw_nafnet_finalize: incomplete encoders." + std::to_string(l) + "." + std::to_string(j));
        for (size_t j = 0; j < n->decoders[l].size(); ++j)
            if (n->decoders[l][j].have != BLOCK_ALL)
                return fail(FW_ERR_INVALID, "fw_nafnet_finalize: incomplete decoders." + std::to_string(l) + "." + std::to_string(j));
    }
    for (size_t j = 0; j < n->middle_blks.size(); ++j)
        if (n->middle_blks[j].have != BLOCK_ALL) return fail(FW_ERR_INVALID, "fw_nafnet_finalize: incomplete middle_blks." + std::to_string(j));
    if (!n->fuse_front && !n->fuse_tail) return FW_OK;
    return guarded([&] {
        DevGuard dg(n->device);
        auto each = [&](Block& bl) {
            if (n->fuse_front && pw_dw_eligible(bl.c, PWDW_GATE_MUL) && !bl.front.p) build_front(n, bl);
            if (n->fuse_tail && bl.c == 128 && !bl.tail128.p) build_tail128(n, bl);
        };
        for (auto& lv : n->encoders) for (Block& bl : lv) each(bl);
        for (auto& lv : n->decoders) for (Block& bl : lv) each(bl);
        for (Block& bl : n->middle_blks) each(bl);
    });
}

int fw_nafnet_denoise_u8(fw_nafnet* n, const uint8_t* in_bgr, int in_loc, int H, int W, uint8_t* out_bgr, int out_loc,
                         float* out_rgb_f32, void* stream) {
    if (!n || !in_bgr) return fail(FW_ERR_INVALID, "fw_nafnet_denoise_u8: NULL argument");
    if (!out_bgr && !out_rgb_f32) return fail(FW_ERR_INVALID, "fw_nafnet_denoise_u8: no output requested");
    if (H < 1 || W < 1 || H > 16384 || W > 16384) return fail(FW_ERR_INVALID, "fw_nafnet_denoise_u8: bad frame size");
    if ((in_loc != FW_HOST && in_loc != FW_DEVICE) || (out_loc != FW_HOST && out_loc != FW_DEVICE))
        return fail(FW_ERR_INVALID, "fw_nafnet_denoise_u8: bad buffer location");
    int rc = fw_nafnet_finalize(n);
    if (rc != FW_OK) return rc;
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        DevGuard dg(n->device);
        hipStream_t st = (hipStream_t)stream;
        StreamOrder::Scope in_order(n->order, st);
        const Plan pl = make_plan(n, H, W);
        if (n->ws.bytes < pl.total) {
            FW_HIP_CHECK(hipDeviceSynchronize());
            drop_graphs(n);
            n->ws.release();
            FW_HIP_CHECK(hipMalloc(&n->ws.p, pl.total));
            n->ws.bytes = pl.total;
        }
        const size_t bytes = (size_t)H * W * 3;
        const uint8_t* d_in = in_bgr;
        if (in_loc == FW_HOST) {
            uint8_t* stg = (uint8_t*)n->ws.p + pl.in_u8;
            FW_HIP_CHECK(hipMemcpyAsync(stg, in_bgr, bytes, hipMemcpyHostToDevice, st));
            d_in = stg;
        }
        uint8_t* d_out = out_bgr;
        if (out_bgr && out_loc == FW_HOST) d_out = (uint8_t*)n->ws.p + pl.out_u8;
        const bool graphed = n->warmed && n->graph_mode == 1;   // the first forward runs eagerly: lazy one-time set-up stays out of a capture
        n->warmed = true;
        if (!graphed) {
            forward(n, d_in, H, W, d_out, out_rgb_f32, st);
        } else {
            fw_nafnet::GraphEntry* hit = nullptr;
            for (auto& g : n->graphs)
                if (g.H == H && g.W == W && g.in == d_in && g.out == d_out && g.rgb == out_rgb_f32) hit = &g;
            if (!hit) {
                if (n->graphs.size() >= 16) drop_graphs(n);
                (void)conv_zero_page();   // its first use allocates: not inside a capture
                hipStream_t cs = nullptr;
                FW_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
                fw_nafnet::GraphEntry e{H, W, d_in, d_out, out_rgb_f32, nullptr, nullptr};
                hipError_t err = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
                if (err == hipSuccess) {
                    try {
                        forward(n, d_in, H, W, d_out, out_rgb_f32, cs);
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

double fw_nafnet_flops(const fw_nafnet* n, int H, int W) {
    if (!n || H < 1 || W < 1) return 0.0;
    const int mult = 1 << n->nlev;
    const double Hp = (H + mult - 1) / mult * mult, Wp = (W + mult - 1) / mult * mult;
    double mac = 0;
    auto block = [](double c) { return c * 2 * c + 2 * c * 9 + c * c + c * 2 * c + c * c; };  // per pixel (SCA mat-vec ignored)
    double px = Hp * Wp;
    mac += px * 9 * 3 * n->width;
    for (int l = 0; l < n->nlev; ++l) {
        const double c = (double)(n->width << l);
        mac += px * block(c) * n->enc[l];
        mac += px / 4 * 4 * c * 2 * c;  // down
        px /= 4;
    }
    mac += px * block((double)(n->width << n->nlev)) * n->middle;
    for (int i = 0; i < n->nlev; ++i) {
        const double chi = (double)(n->width << (n->nlev - i));
        mac += px * chi * 2 * chi;  // up 1x1
        px *= 4;
        mac += px * block(chi / 2) * n->dec[i];
    }
    mac += px * 9 * n->width * 3;
    return 2.0 * mac;
}

// ---- K8: blend kernels of the TAP driver (tap_denoise.py:417-534, :614-618) ---------------------------------------
int fw_u8_crop(const uint8_t* src, int height, int width, int y0, int x0, int th, int tw, uint8_t* dst, void* stream) {
    if (!src || !dst || th < 1 || tw < 1 || y0 < 0 || x0 < 0 || y0 + th > height || x0 + tw > width)
        return fail(FW_ERR_INVALID, "fw_u8_crop: window outside the frame");
    return guarded([&] { launch_u8_crop(src, width, y0, x0, th, tw, dst, (hipStream_t)stream); });
}

int fw_tile_blend_accumulate(float* acc, float* wsum, int height, int width, const uint8_t* tile, int y0, int x0, int th,
                             int tw, int overlap, void* stream) {
    if (!acc || !wsum || !tile || th < 1 || tw < 1 || y0 < 0 || x0 < 0 || y0 + th > height || x0 + tw > width || overlap < 0 ||
        overlap > th || overlap > tw)
        return fail(FW_ERR_INVALID, "fw_tile_blend_accumulate: bad tile geometry");
    return guarded([&] {
        launch_tile_blend_acc(acc, wsum, width, tile, y0, x0, th, tw, overlap, y0 > 0, y0 + th < height, x0 > 0,
                              x0 + tw < width, (hipStream_t)stream);
    });
}

int fw_tile_blend_finish(const float* acc, const float* wsum, int height, int width, uint8_t* out, void* stream) {
    if (!acc || !wsum || !out || height < 1 || width < 1) return fail(FW_ERR_INVALID, "fw_tile_blend_finish: bad argument");
    return guarded([&] { launch_tile_blend_finish(acc, wsum, (long)height * width, out, (hipStream_t)stream); });
}

int fw_temporal_average_u8(const uint8_t* const* frames, const float* weights, int count, size_t nbytes, uint8_t* out,
                           void* stream) {
    if (!frames || !weights || !out || count < 1 || count > 16) return fail(FW_ERR_INVALID, "fw_temporal_average_u8: bad argument");
    for (int k = 0; k < count; ++k)
        if (!frames[k]) return fail(FW_ERR_INVALID, "fw_temporal_average_u8: NULL frame");
    return guarded([&] { launch_temporal_average(frames, weights, count, (long)nbytes, out, (hipStream_t)stream); });
}

int fw_strength_blend_u8(const uint8_t* original, const uint8_t* denoised, double strength, size_t nbytes, uint8_t* out,
                         void* stream) {
    if (!original || !denoised || !out || !(strength >= 0.0 && strength <= 1.0))
        return fail(FW_ERR_INVALID, "fw_strength_blend_u8: bad argument");
    // (1 - s) is formed in double like the reference's Python float arithmetic, then rounded once to float32
    const float oms = (float)(1.0 - strength);
    const float sf = (float)strength;
    return guarded([&] { launch_strength_blend(original, denoised, oms, sf, (long)nbytes, out, (hipStream_t)stream); });
}

int fw_grain_addback_u8(const uint8_t* original, const uint8_t* denoised, int height, int width, double factor,
                        uint16_t* scratch, uint8_t* out, void* stream) {
    if (!original || !denoised || !scratch || !out || height < 1 || width < 1 || !(factor >= 0.0) || factor > 1.0)
        return fail(FW_ERR_INVALID, "fw_grain_addback_u8: bad argument");
    return guarded([&] { launch_grain_addback(original, denoised, height, width, factor, scratch, out, (hipStream_t)stream); });
}

int fw_resize_lanczos4_u8(const uint8_t* src, int src_h, int src_w, int channels, uint8_t* dst, int dst_h, int dst_w, void* stream) {
    if (!src || !dst || src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1 || channels < 1 || channels > 4)
        return fail(FW_ERR_INVALID, "fw_resize_lanczos4_u8: bad argument");
    return guarded([&] { launch_resize_lanczos4_u8(src, src_h, src_w, channels, dst, dst_h, dst_w, (hipStream_t)stream); });
}

int fw_resize_lanczos4_u16(const uint16_t* src, int src_h, int src_w, int channels, uint16_t* dst, int dst_h, int dst_w, void* stream) {
    if (!src || !dst || src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1 || channels < 1 || channels > 4)
        return fail(FW_ERR_INVALID, "fw_resize_lanczos4_u16: bad argument");
    return guarded([&] { launch_resize_lanczos4_u16(src, src_h, src_w, channels, dst, dst_h, dst_w, (hipStream_t)stream); });
}

int fw_resize_linear_u8(const uint8_t* src, int src_h, int src_w, int channels, uint8_t* dst, int dst_h, int dst_w, void* stream) {
    if (!src || !dst || src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1 || channels < 1 || channels > 4)
        return fail(FW_ERR_INVALID, "fw_resize_linear_u8: bad argument");
    return guarded([&] { launch_resize_linear_u8(src, src_h, src_w, channels, dst, dst_h, dst_w, (hipStream_t)stream); });
}

int fw_face_paste_u8(uint8_t* frame, int height, int width, int x1, int y1, int x2, int y2, const uint8_t* enhanced, double strength,
                     void* stream) {
    if (!frame || !enhanced || height < 1 || width < 1 || !(strength >= 0.0) || strength > 1.0)
        return fail(FW_ERR_INVALID, "fw_face_paste_u8: bad argument");
    return guarded([&] { launch_face_paste_u8(frame, height, width, x1, y1, x2, y2, enhanced, (float)strength, (hipStream_t)stream); });
}

int fw_nafnet_destroy(fw_nafnet* n) {
    if (!n) return FW_OK;
    { std::lock_guard<std::mutex> lk(n->mu); }   // a call in flight on another thread finishes first
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(n->device);
    (void)hipDeviceSynchronize();
    drop_graphs(n);
    auto free_block = [](Block& b) {
        for (DevBuf* d : {&b.n1w, &b.n1b, &b.n2w, &b.n2b, &b.beta, &b.gamma, &b.w1, &b.b1, &b.w3, &b.b3, &b.w4, &b.b4, &b.w5, &b.w1g, &b.w3g, &b.w4g, &b.w5g,
                          &b.b5, &b.wdw, &b.bdw, &b.wsca, &b.bsca, &b.front, &b.tail128})
            d->release();
    };
    for (auto& v : n->encoders) for (auto& b : v) free_block(b);
    for (auto& v : n->decoders) for (auto& b : v) free_block(b);
    for (auto& b : n->middle_blks) free_block(b);
    for (auto& l : n->downs) { l.w.release(); l.b.release(); }
    for (auto& l : n->ups) { l.w.release(); l.b.release(); }
    n->intro_w.release(); n->intro_b.release(); n->ending_w.release(); n->ending_b.release();
    n->ws.release();
    if (prev >= 0) (void)hipSetDevice(prev);
    n->order.destroy();
    delete n;
    return FW_OK;
}

size_t fw_preserve_edges_scratch_bytes(int height, int width) {
    if (height < 1 || width < 1) return 0;
    return preserve_edges_scratch_bytes(height, width);
}

int fw_preserve_edges_u8(const uint8_t* original, const uint8_t* denoised, int height, int width, double low_threshold,
                         double high_threshold, void* scratch, uint8_t* out, void* stream) {
    if (!original || !denoised || !scratch || !out || height < 1 || width < 1 || !(low_threshold >= 0) || !(high_threshold >= 0))
        return fail(FW_ERR_INVALID, "fw_preserve_edges_u8: bad argument");
    return guarded([&] {
        double lo = low_threshold, hi = high_threshold;
        if (lo > hi) std::swap(lo, hi);
        launch_preserve_edges(original, denoised, height, width, (int)floor(lo), (int)floor(hi), scratch, out, (hipStream_t)stream);
    });
}

}  // extern "C"
