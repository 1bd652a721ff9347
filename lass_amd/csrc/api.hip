// api.hip - the C-ABI of include/lass_hip.h: context, weight ingestion/folding, workspace plan, forward orchestration.
//
// Forward order mirrors /root/reference/models/resunet.py:522-595 (ResUNet30_Base.forward) with FiLM (:59-81) hoisted to
// one launch.  torch.cat of the decoder (:258) is virtual: encoder blocks write their skip output straight into the
// second channel half of the decoder's concat buffer and the transposed conv writes the first half.
#include <hip/hip_runtime.h>
#include <atomic>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/lass_hip.h"
#include "kernels.h"

namespace {

struct EncSpec { const char* name; int cin, cout, dh, dw; };
struct DecSpec { const char* name; int cin, cout, uh, uw; };
// resunet.py:315-418
const EncSpec kEnc[7] = {{"encoder_block1", 32, 32, 2, 2},   {"encoder_block2", 32, 64, 2, 2},
                         {"encoder_block3", 64, 128, 2, 2},  {"encoder_block4", 128, 256, 2, 2},
                         {"encoder_block5", 256, 384, 2, 2}, {"encoder_block6", 384, 384, 1, 2},
                         {"conv_block7a", 384, 384, 1, 1}};
const DecSpec kDec[6] = {{"decoder_block1", 384, 384, 1, 2}, {"decoder_block2", 384, 384, 2, 2},
                         {"decoder_block3", 384, 256, 2, 2}, {"decoder_block4", 256, 128, 2, 2},
                         {"decoder_block5", 128, 64, 2, 2},  {"decoder_block6", 64, 32, 2, 2}};
constexpr int kPreCh = 32;
constexpr float kBnEps = 1e-5f;
constexpr int kMaxBranches = LASS_MAX_STFT_WINDOWS;

struct Raw {
    float* d = nullptr;
    std::vector<int64_t> shape;
    size_t n = 0;
};

struct Site {
    std::string film;  // "encoder_block1->conv_block1->beta1"
    std::string bn;    // "base.encoder_block1.conv_block1.bn1"
    int C = 0;
    int off = 0;
};

struct ResBlock {  // one ConvBlockRes
    std::string prefix;  // "base.encoder_block1.conv_block1"
    int cin = 0, cout = 0;
    int width = 0;         // bins of the level the block runs at (fcrop >> level): known at finalize, the frame count is not
    int s1 = -1, s2 = -1;  // site indices
    float *w1 = nullptr, *w2 = nullptr, *wsc = nullptr;  // re-laid-out
    float *u1 = nullptr, *u2 = nullptr, *usc = nullptr;  // Winograd-domain copies (when enabled)
    float *u1r = nullptr, *u2r = nullptr, *uscr = nullptr;  // 32-cout blocks: resident LDS images of wino32.hip
    float *u1f = nullptr, *u2f = nullptr;                   // conv1 / conv2 in the F(4x4,3x3) domain (wino4.hip), deep-K blocks
    void *b1 = nullptr, *b2 = nullptr, *bsc16 = nullptr;  // bf16 copies (LASS_COMPUTE_BF16 / _BF16X3)
    void *b1l = nullptr, *b2l = nullptr, *bscl = nullptr;  // lo halves of the hi+lo split (LASS_COMPUTE_BF16X3)
    const float* bsc = nullptr;                         // raw shortcut bias
};

struct ProfEntry {
    const char* name;
    double ms = 0;
    int launches = 0;
};

thread_local std::string g_create_err;

}  // namespace

// Model geometry.  variant 0 = ResUNet30 (models/resunet.py): one analysis branch, n_fft = win = 1024.
// variant 1 = the multi-resolution-STFT separator (models/resunet_with_multistft.py under the authored spec of
// DESIGN.md / lass_amd/arch.py): nbr analysis windows at a common n_fft, one pre_conv + encoder_block1 per window,
// channel-concatenated pools / skips, shared trunk, mask and iSTFT on the branch `mask_br`.
struct Geometry {
    int variant = 0;
    int nbr = 1;
    int wins[kMaxBranches] = {1024, 0, 0, 0};
    int nfft = 1024, nbins = 513, fcrop = 512;
    int mask_br = 0;
    int magphase_sem = 0;  // 0: base.py:83-88 clamp on |X|^2; 1: torchlibrosa magphase (precomputed-STFT wire format)
};

struct lass_ctx {
    int device = 0;
    std::string err;
    Geometry g;
    EncSpec E[7];  // encoder table of THIS model: E[0] is one analysis branch's block; E[1].cin = 32 * nbr
    DecSpec D[6];
    int dec_cat[6] = {0};              // concat channels of decoder d = D[d].cout + skip channels
    std::string pre_name[kMaxBranches];  // "base.pre_conv" / "base.pre_convs.<win>"
    std::map<std::string, Raw> raw;
    bool finalized = false;
    float2* tw2k = nullptr;  // (cos, sin)(2 pi k / 2048): twiddles of every transform size + the Hann windows
    std::vector<Site> sites;
    std::map<std::string, int> site_idx;
    int n_shift = 0;
    float *film_W = nullptr, *film_b = nullptr, *bn_scale = nullptr, *bn_base = nullptr;
    float *bn0_s = nullptr, *bn0_h = nullptr;
    std::vector<ResBlock> enc, dec;  // enc: nbr branch blocks (encoder_block1[s]) then the 6 trunk blocks; dec: 6
    int dec_site[6] = {0};           // decoder_blockN->beta1
    void* up16[6] = {nullptr};       // bf16 transposed-conv weights (hi) per decoder, bf16 modes only
    void* up16l[6] = {nullptr};      // lo halves (LASS_COMPUTE_BF16X3)
    std::vector<void*> owned;        // derived device buffers to free
    // profiling
    int compute_mode = LASS_COMPUTE_F32;
    bool wino = true;          // Winograd F(2x2,3x3) kernels for the 3x3 convs at W >= 32 (LASS_WINO=0: direct only)
    int wino4_mincin = 32;     // 3x3 convs with at least that many input channels (and >= 32-wide images) run as Winograd
                               // F(4x4,3x3) (wino4.hip); LASS_WINO4=<min Cin>, 0 = off (F(2x2,3x3) everywhere)
    bool wino32 = true;        // weights-resident persistent kernel for the 32-cout layers (LASS_WINO32=0: wino.hip everywhere)
    int live_class = 0;        // what this finalized context counts as in the process-wide packed-f32 guard (0 none, 1 bf16 MFMA
                               // kernels, 2 routes launches to wino32.hip): see packed_guard_enter
    bool fuse_preconv = true;  // LASS_FUSE_PRECONV=0 materialises pre_conv's output with its own kernel
    bool fuse_pool = true;  // LASS_FUSE_POOL=0 selects the stand-alone pool kernel (A/B + parity of both paths)
    bool fuse_catb = true;  // bf16 mode: decoder concats as blocked bf16 copies (LASS_FUSE_CATB=0: f32 concat)
    bool fuse_block = true;  // bf16 mode: encoder_block1 as one kernel, intermediate in LDS (LASS_FUSE_BLOCK=0: two launches)
    bool fuse_up = true;    // bf16 mode: decoder_block6's transposed conv inside its fused kernel (LASS_FUSE_UP=0: its own launch)
    void* up_sc16 = nullptr;  // ... the 1x1 shortcut composed with that transposed conv, bf16 [4][2][128] units (lass_finalize)
    bool fuse_mask = true;  // LASS_FUSE_MASK=0 keeps after_conv + mask as their own kernel behind decoder_block6
    // hipGraph replay of lass_separate (LASS_GRAPH=0 disables): the ~40 launches of one (pointers, shape) combination are
    // captured once on an internal stream and replayed on the caller's stream
    bool use_graph = true;
    struct GraphKey {
        const void *mix = nullptr, *cond = nullptr, *out = nullptr, *ws = nullptr;
        int B = 0, L = 0;
        unsigned long gen = 0;
        bool operator==(const GraphKey& o) const {
            return mix == o.mix && cond == o.cond && out == o.out && ws == o.ws && B == o.B && L == o.L && gen == o.gen;
        }
    };
    // A small cache of instantiated graphs: the evaluator alternates between its common batch and a ragged tail, long-form
    // callers between a few window counts.  An entry is captured on the third call that presents its key.
    struct GraphEntry {
        GraphKey key;
        hipGraphExec_t exec = nullptr;
        size_t need = 0;         // workspace bytes the captured plan addresses (re-checked on every replay)
        int seen = 0;            // calls with this key so far
        unsigned long used = 0;  // g_tick of the last call (LRU)
        bool split = false;      // the captured launch sequence runs part-batches (their layouts are what the workspace holds)
    };
    // (B, L) -> did the LAST lass_separate of that shape run as part-batches?  lass_workspace_tensor refuses only then: eager
    // calls (the first calls of a key, LASS_GRAPH=0, changing pointers) leave the whole-batch layout, taps stay readable.
    std::map<std::pair<int, int>, bool> last_split;
    static constexpr int kGraphSlots = 4;
    GraphEntry g_slots[kGraphSlots];
    unsigned long g_tick = 0;
    std::vector<hipGraphExec_t> g_retired;  // replaced execs: a replay may still be in flight on some stream, so they are
                                            // destroyed only behind a device synchronisation (lass_finalize / lass_destroy)
    hipStream_t g_stream = nullptr;
    // Half-batch overlap: an even batch of >= 8 clips runs as two independent half-batches on two streams (clips are
    // independent: eval-mode BN) - the second on `s2`, forked from / joined to the caller's stream by events - so that one
    // half's small launches (the 16-/8-bin layers: a few hundred workgroups) and launch tails run beside the other half's
    // full-size launches.  Same kernels, same per-clip arithmetic (bit-identical: batch invariance), same workspace size.
    // Measured: bf16 +3.4 %, f32 +0.5 ... +2.1 % depending on the box, split-bf16 +1.1 %.
    // 1 (default): only inside the captured hipGraph that lass_separate replays - an EAGER two-stream launch depends on the
    // process having a free hardware queue for `s2` (with an RCCL communicator alive in the process it measured 8 % SLOWER than
    // the unsplit launch, without one 2 % faster; a graph's branches do not care).  LASS_SPLIT=2: eager launches too; 0: never.
    // DESIGN.md section 5b has the measurements and the co-residency hazard found on the way.
    int split_batch = 1;
    int split_parts = 2;  // LASS_SPLIT_PARTS: 2 or 4 part-batches (each of at least 4 clips)
    hipStream_t s2[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    unsigned long gen = 0;         // bumped by lass_finalize: a graph holds weight pointers
    long g_replays = 0, g_captures = 0;
    bool profiling = false;
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> ev_pool;
    struct Pending { int cls; hipEvent_t a, b; };
    std::vector<Pending> pending;
    size_t ev_used = 0;
};

namespace {

#define HIP_TRY(ctx, expr)                                                                                     \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess) {                                                                                \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                                     \
            return LASS_ERR_HIP;                                                                               \
        }                                                                                                      \
    } while (0)

int fail(lass_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

bool starts_with(const std::string& s, const char* p) { return s.rfind(p, 0) == 0; }
bool ends_with(const std::string& s, const char* p) {
    const size_t n = strlen(p);
    return s.size() >= n && s.compare(s.size() - n, n, p) == 0;
}

std::string film_to_bn(const std::string& film) {  // 'a->b->beta1' -> 'base.a.b.bn1'
    std::string s = "base.";
    for (size_t i = 0; i < film.size(); ++i) {
        if (film[i] == '-' && i + 1 < film.size() && film[i + 1] == '>') {
            s += '.';
            ++i;
        } else {
            s += film[i];
        }
    }
    const size_t p = s.rfind("beta");
    s.replace(p, 4, "bn");
    return s;
}

int add_site(lass_ctx* c, const std::string& film, int C) {
    Site s;
    s.film = film;
    s.bn = film_to_bn(film);
    s.C = C;
    s.off = c->n_shift;
    c->n_shift += C;
    c->site_idx[film] = (int)c->sites.size();
    c->sites.push_back(s);
    return (int)c->sites.size() - 1;
}

const ResBlock& trunk_block(const lass_ctx* c, int i) { return c->enc[c->g.nbr - 1 + i]; }  // i = 1..6 (E[i])

void build_arch(lass_ctx* c) {
    const Geometry& g = c->g;
    c->sites.clear();
    c->site_idx.clear();
    c->n_shift = 0;
    c->enc.clear();
    c->dec.clear();
    for (int i = 0; i < 7; ++i) c->E[i] = kEnc[i];
    for (int i = 0; i < 6; ++i) c->D[i] = kDec[i];
    c->E[1].cin = kPreCh * g.nbr;
    for (int k = 0; k < g.nbr; ++k) {  // analysis branches
        ResBlock rb;
        std::string film;
        if (g.variant == 0) {
            rb.prefix = "base.encoder_block1.conv_block1";
            film = "encoder_block1->conv_block1";
            c->pre_name[k] = "base.pre_conv";
        } else {
            const std::string w = std::to_string(g.wins[k]);
            rb.prefix = "base.encoder_block1s." + w + ".conv_block1";
            film = "encoder_block1s->" + w + "->conv_block1";
            c->pre_name[k] = "base.pre_convs." + w;
        }
        rb.cin = rb.cout = kPreCh;
        rb.width = g.fcrop;
        rb.s1 = add_site(c, film + "->beta1", kPreCh);
        rb.s2 = add_site(c, film + "->beta2", kPreCh);
        c->enc.push_back(rb);
    }
    for (int i = 1; i < 7; ++i) {
        const EncSpec& e = c->E[i];
        ResBlock rb;
        rb.prefix = std::string("base.") + e.name + ".conv_block1";
        rb.cin = e.cin;
        rb.cout = e.cout;
        rb.width = g.fcrop >> i;  // encoder i (0-based) runs behind i frequency halvings: 512, 256, ..., 16, 8 bins
        rb.s1 = add_site(c, std::string(e.name) + "->conv_block1->beta1", e.cin);
        rb.s2 = add_site(c, std::string(e.name) + "->conv_block1->beta2", e.cout);
        c->enc.push_back(rb);
    }
    for (int i = 0; i < 6; ++i) {
        const auto& d = c->D[i];
        const int e = 5 - i;
        c->dec_cat[i] = d.cout + c->E[e].cout * (e == 0 ? g.nbr : 1);  // torch.cat((x, skip), 1)
        c->dec_site[i] = add_site(c, std::string(d.name) + "->beta1", d.cin);
        ResBlock rb;
        rb.prefix = std::string("base.") + d.name + ".conv_block2";
        rb.cin = c->dec_cat[i];
        rb.cout = d.cout;
        rb.width = g.fcrop >> e;  // decoder i runs at its skip's level: 16, 32, ..., 512 bins
        rb.s1 = add_site(c, std::string(d.name) + "->conv_block2->beta1", rb.cin);
        rb.s2 = add_site(c, std::string(d.name) + "->conv_block2->beta2", d.cout);
        c->dec.push_back(rb);
    }
}

// Expected shape of a required parameter, or empty if the name is not one.
std::vector<int64_t> expected_shape(const lass_ctx* c, const std::string& name) {
    auto bn_field = [](const std::string& n) {
        return ends_with(n, ".weight") || ends_with(n, ".bias") || ends_with(n, ".running_mean") ||
               ends_with(n, ".running_var");
    };
    if (starts_with(name, "base.bn0.") && bn_field(name)) return {c->g.nbins};
    for (int k = 0; k < c->g.nbr; ++k) {
        if (name == c->pre_name[k] + ".weight") return {kPreCh, 1, 1, 1};
        if (name == c->pre_name[k] + ".bias") return {kPreCh};
    }
    if (name == "base.after_conv.weight") return {3, kPreCh, 1, 1};
    if (name == "base.after_conv.bias") return {3};
    auto res_block = [&](const ResBlock& rb) -> std::vector<int64_t> {
        const std::string& p = rb.prefix;
        if (!starts_with(name, (p + ".").c_str())) return {};
        const std::string f = name.substr(p.size() + 1);
        if (starts_with(f, "bn1.") && bn_field(f)) return {rb.cin};
        if (starts_with(f, "bn2.") && bn_field(f)) return {rb.cout};
        if (f == "conv1.weight") return {rb.cout, rb.cin, 3, 3};
        if (f == "conv2.weight") return {rb.cout, rb.cout, 3, 3};
        if (rb.cin != rb.cout && f == "shortcut.weight") return {rb.cout, rb.cin, 1, 1};
        if (rb.cin != rb.cout && f == "shortcut.bias") return {rb.cout};
        return {};
    };
    for (const auto& rb : c->enc) {
        auto s = res_block(rb);
        if (!s.empty()) return s;
    }
    for (int i = 0; i < 6; ++i) {
        auto s = res_block(c->dec[i]);
        if (!s.empty()) return s;
        const DecSpec& d = c->D[i];
        const std::string p = std::string("base.") + d.name;
        if (name == p + ".conv1.weight") return {d.cin, d.cout, d.uh, d.uw};
        if (starts_with(name, (p + ".bn1.").c_str()) && bn_field(name)) return {d.cin};
    }
    if (starts_with(name, "film.")) {
        for (const auto& s : c->sites) {
            if (name == "film." + s.film + ".weight") return {s.C, LASS_COND};
            if (name == "film." + s.film + ".bias") return {s.C};
        }
    }
    return {};
}

bool ignorable(const std::string& name) {
    if (starts_with(name, "base.stft.") || starts_with(name, "base.istft.")) return true;
    if (ends_with(name, "num_batches_tracked")) return true;
    for (const auto& d : kDec) {
        if (starts_with(name, (std::string("base.") + d.name + ".bn2.").c_str())) return true;      // resunet.py:230
        if (starts_with(name, (std::string("film.") + d.name + "->beta2.").c_str())) return true;  // never read
    }
    return false;
}

const float* rawp(const lass_ctx* c, const std::string& name) {
    auto it = c->raw.find(name);
    return it == c->raw.end() ? nullptr : it->second.d;
}

template <typename T>
int dev_alloc(lass_ctx* c, T** p, size_t count) {
    void* v = nullptr;
    HIP_TRY(c, hipMalloc(&v, count * sizeof(T)));
    c->owned.push_back(v);
    *p = (T*)v;
    return 0;
}

void free_owned(lass_ctx* c) {
    for (void* p : c->owned) (void)hipFree(p);
    c->owned.clear();
}

// ---- profiling ----------------------------------------------------------------------------------------------------
enum ProfClass { P_STFT = 0, P_FILM, P_PRECONV, P_CONV3X3, P_TCONV, P_POOL, P_MASK, P_ISTFT, P_COUNT };
const char* kProfNames[P_COUNT] = {"stft_magphase", "film", "pre_conv", "conv3x3_mfma", "tconv_mfma",
                                   "avg_pool",      "mask_apply", "istft"};

// Event pairs come from a pool that lass_set_profiling sizes up front: nothing is created (or allocated) inside
// lass_separate.  When a caller lets more than kProfPairs scopes accumulate without collecting them
// (lass_profile_get / lass_profile_reset), the surplus scopes are simply not timed.
constexpr size_t kProfPairs = 512;

struct ProfScope {
    lass_ctx* c;
    hipStream_t s;
    int cls;
    hipEvent_t b = nullptr;
    bool on = false;
    ProfScope(lass_ctx* ctx, hipStream_t st, int k) : c(ctx), s(st), cls(k) {
        if (!c->profiling || c->ev_used + 2 > c->ev_pool.size()) return;
        hipEvent_t a = c->ev_pool[c->ev_used++];
        b = c->ev_pool[c->ev_used++];
        (void)hipEventRecord(a, s);
        c->pending.push_back({cls, a, b});
        on = true;
    }
    ~ProfScope() {
        if (on) (void)hipEventRecord(b, s);
    }
};

void prof_collect(lass_ctx* c) {
    for (auto& p : c->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->prof[p.cls].ms += ms;
            c->prof[p.cls].launches += 1;
        }
    }
    c->pending.clear();
    c->ev_used = 0;
}

// ---- one residual block ---------------------------------------------------------------------------------------------
// x: (B,cin,H,W) batch stride x_bs; out: batch stride out_bs (may be a channel slice of a concat buffer).
// pool_out (optional): the block's avg-pooled output (B,cout,H/pool_h,W/2), produced by conv2's epilogue.
// x0 (optional, encoder_block1 only): the block input is pre_conv(x0) and is formed on the fly - x is then ignored.
// bf16 mode: a decoder's concat input (transposed-conv output | encoder skip) kept as two blocked bf16 copies in the
// concat buffer's storage - `act` (consumer prologue applied) for conv1, `raw` for the 1x1 shortcut (ConvArgs::out_bf16_act)
struct CatCopies {
    void* act;
    void* raw;
    int noct;            // octets per clip = 2C / 8
    const float* scale;  // conv_block2.bn1 (+ FiLM shift below) of the decoder, at concat channel 0
    const float* shift;
};

// Output head fused into the last decoder block's conv2 (after_conv + complex ratio mask): inputs / outputs of the mask
struct MaskHead {
    const float* mag;
    const float* cosv;
    const float* sinv;
    float* oreal;
    float* oimag;
    int T;
    int nbins;
};

// encoder_block1[s] input: the block input is pre_conv(x0) (1 -> 32 channels, resunet.py:555), formed while staging
struct PreConv {
    const float* x0;
    const float* w;
    const float* b;
};

// bf16 mode, decoder_block6: the transposed conv in front of the block runs inside the fused kernel (conv_bf16_fused.hip)
struct UpFuse {
    const void* x_act;   // previous decoder's output, that conv's prologue applied, blocked bf16 (B, cin/8, h, w, 8)
    int cin, h, w;
    const void* w16;     // transposed-conv weights, bf16
    const void* wsc16;   // shortcut (up-sampled half) composed with the transposed conv, bf16
};

int run_resblock(lass_ctx* c, const ResBlock& rb, const float* x, long x_bs, int B, int H, int W, const float* shift,
                 float* a2, float* out, long out_bs, hipStream_t st, float* pool_out = nullptr, int pool_h = 2,
                 const PreConv* pre = nullptr, const MaskHead* mh = nullptr, const CatCopies* skip_out = nullptr,
                 const CatCopies* cat_in = nullptr, const Site* act_out = nullptr, const CatCopies* pool_copies = nullptr,
                 long pool_bs = 0, const UpFuse* up = nullptr, int cofs = 0) {  // cofs: this branch's channel offset inside the concatenated skip / pool
    const float* x0 = pre ? pre->x0 : nullptr;
    const Site& s1 = c->sites[rb.s1];
    const Site& s2 = c->sites[rb.s2];
    const long HW = (long)H * W;
    ConvArgs p;
    p.in = x; p.in_bs = x_bs; p.Cin = rb.cin; p.w = rb.w1; p.Nw = rb.cout; p.N = rb.cout;
    p.pro_scale = c->bn_scale + s1.off; p.pro_shift = shift + s1.off; p.pro_shift_bs = c->n_shift;
    p.epi_scale = c->bn_scale + s2.off; p.epi_shift = shift + s2.off; p.epi_shift_bs = c->n_shift;
    p.out = a2; p.out_bs = rb.cout * HW; p.B = B; p.H = H; p.W = W;
    if (x0) {
        p.in = x0; p.in_bs = HW;
        p.pre_w = pre->w; p.pre_b = pre->b;
    }
    p.w_wino = rb.u1; p.w_wino32 = rb.u1r; p.w_wino4 = rb.u1f;
    p.w_bf16 = rb.b1; p.w_bf16_lo = rb.b1l;
    const bool bf1 = c->compute_mode != LASS_COMPUTE_F32 && rb.b1 && rb.b2 && lass_bf16_supported(p) &&
                     (!x0 || W % 32 == 0) && rb.cout % 16 == 0 && (rb.cin == rb.cout || (rb.bsc16 && rb.cin % 16 == 0));
    // bf16 modes: the intermediate a2 is kept as blocked bf16 (hi, and lo for the split mode) in the same scratch
    void* a2_hi = a2;
    void* a2_lo = c->compute_mode == LASS_COMPUTE_BF16X3 ? (void*)((char*)a2 + (size_t)B * rb.cout * HW * 2) : nullptr;
    if (bf1) { p.out_bf16 = a2_hi; p.out_bf16_lo = a2_lo; }
    if ((skip_out || cat_in) && (!bf1 || c->compute_mode != LASS_COMPUTE_BF16))
        return fail(c, LASS_ERR_STATE, "blocked bf16 concat copies need the bf16 kernels");
    if (cat_in) p.in_bf16 = cat_in->act;
    const bool wino1 = !bf1 && c->wino && rb.u1 && lass_wino_supported(p);
    auto launch_conv1 = [&]() -> int {
        ProfScope ps(c, st, P_CONV3X3);
        if (bf1)
            HIP_TRY(c, lass_launch_conv_bf16(x0 ? CONV1_ACT_PRE : CONV1_ACT, p, st));
        else if (wino1 && rb.u1f && lass_wino4_supported(x0 ? CONV1_ACT_PRE : CONV1_ACT, p))
            HIP_TRY(c, lass_launch_wino4(x0 ? CONV1_ACT_PRE : CONV1_ACT, p, st));
        else if (wino1 && c->wino32 && lass_wino32_supported(x0 ? CONV1_ACT_PRE : CONV1_ACT, p))
            HIP_TRY(c, lass_launch_wino32(x0 ? CONV1_ACT_PRE : CONV1_ACT, p, st));
        else if (wino1)
            HIP_TRY(c, lass_launch_wino(x0 ? CONV1_ACT_PRE : CONV1_ACT, p, st));
        else
            HIP_TRY(c, lass_launch_conv(x0 ? CONV1_ACT_PRE : CONV1_ACT, p, st));
        return 0;
    };
    ConvArgs q;
    q.in = a2; q.in_bs = rb.cout * HW; q.Cin = rb.cout; q.w = rb.w2; q.Nw = rb.cout; q.N = rb.cout;
    q.out = out; q.out_bs = out_bs; q.B = B; q.H = H; q.W = W;
    q.pool_out = pool_out; q.pool_h = pool_h; q.pool_bs = pool_bs;
    q.w_wino = rb.u2; q.w2_wino = rb.usc; q.w_wino32 = rb.u2r; q.w2_wino32 = rb.uscr; q.w_wino4 = rb.u2f;
    if (mh) {  // the block output is consumed by the fused head and never written
        q.out = nullptr;
        q.mask_w = rawp(c, "base.after_conv.weight"); q.mask_b = rawp(c, "base.after_conv.bias");
        q.mask_mag = mh->mag; q.mask_cos = mh->cosv; q.mask_sin = mh->sinv;
        q.mask_re = mh->oreal; q.mask_im = mh->oimag; q.mask_T = mh->T; q.mask_nbins = mh->nbins;
    }
    q.w_bf16 = rb.b2; q.w2_bf16 = rb.bsc16; q.w_bf16_lo = rb.b2l; q.w2_bf16_lo = rb.bscl;
    const bool bf2 = bf1;  // conv1 and conv2 of a block share shape and mode: both or neither
    if (bf2) { q.in_bf16 = a2_hi; q.in_bf16_lo = a2_lo; }
    if (cat_in) q.in2_bf16 = cat_in->raw;
    if (pool_copies) {  // bf16 mode: the pooled output as blocked bf16 copies for the next encoder block
        if (!bf2 || c->compute_mode != LASS_COMPUTE_BF16 || !pool_out || pool_h != 2)
            return fail(c, LASS_ERR_STATE, "blocked bf16 pooled copies need the bf16 kernels and the fused 2x2 pool");
        q.pool_out = nullptr;
        q.pool_bf16 = pool_copies->raw; q.pool_bf16_act = pool_copies->act;
        q.pool_oct0 = cofs / 8; q.pool_noct = pool_copies->noct;
        q.pool_act_scale = pool_copies->scale + cofs; q.pool_act_shift = pool_copies->shift + cofs; q.act_shift_bs = c->n_shift;
    }
    if (act_out) {  // bf16 mode: the block output goes to the next transposed conv only - written as ONE blocked bf16
                    // tensor with that conv's BN+FiLM+leaky prologue already applied (in `out`'s storage)
        if (!bf2 || c->compute_mode != LASS_COMPUTE_BF16) return fail(c, LASS_ERR_STATE, "activated bf16 output needs the bf16 kernels");
        q.out_bf16 = out; q.out = nullptr; q.out_oct0 = 0; q.out_noct = 0;
        q.epi_scale = c->bn_scale + act_out->off; q.epi_shift = shift + act_out->off; q.epi_shift_bs = c->n_shift;
    }
    if (skip_out) {  // the skip goes out as the two blocked copies (concat channels [C, 2C)) instead of f32
        q.out = nullptr;
        q.out_bf16 = skip_out->raw; q.out_bf16_act = skip_out->act;
        q.out_oct0 = (rb.cout + cofs) / 8; q.out_noct = skip_out->noct;
        q.act_scale = skip_out->scale + rb.cout + cofs; q.act_shift = skip_out->shift + rb.cout + cofs; q.act_shift_bs = c->n_shift;
    }
    const bool wino2 = !bf2 && c->wino && rb.u2 && lass_wino_supported(q);
    if (rb.cin == rb.cout) {
        q.res = x; q.res_bs = x_bs;
        if (x0) {
            q.res = x0; q.res_bs = HW;
            q.pre_w = pre->w; q.pre_b = pre->b;
        }
    }
    // bf16 mode, encoder_block1 in the blocked-copy pipeline: the whole block as ONE kernel, its 32-channel intermediate
    // kept in LDS (conv_bf16_fused.hip; LASS_FUSE_BLOCK=0 restores the two launches)
    if (bf2 && x0 && c->fuse_block && c->compute_mode == LASS_COMPUTE_BF16 && skip_out && rb.cin == rb.cout &&
        lass_enc1_fused_bf16_supported(p, q)) {
        ProfScope ps(c, st, P_CONV3X3);
        HIP_TRY(c, lass_launch_enc1_fused_bf16(p, q, st));
        return 0;
    }
    if (rb.cin != rb.cout) { q.in2 = x; q.in2_bs = x_bs; q.Cin2 = rb.cin; q.w2 = rb.wsc; q.bias = rb.bsc; }
    // ... and decoder_block6's ConvBlockRes with the output head behind it (conv1 from the activated cat copy, the 1x1
    // shortcut from the raw one)
    if (up) {  // the caller has NOT run the transposed conv: only the kernel that contains it will do
        ConvArgs uq;
        uq.in_bf16 = up->x_act; uq.Cin = up->cin; uq.H = up->h; uq.W = up->w; uq.B = B;
        uq.w_bf16 = up->w16; uq.w2_bf16 = up->wsc16;
        if (!(bf2 && !x0 && c->fuse_block && c->compute_mode == LASS_COMPUTE_BF16 && cat_in && mh && rb.cin != rb.cout &&
              lass_dec6u_fused_bf16_supported(p, q, uq)))
            return fail(c, LASS_ERR_STATE, "decoder_block6 with its transposed conv inside needs the fused bf16 kernel");
        ProfScope ps(c, st, P_CONV3X3);
        HIP_TRY(c, lass_launch_dec6u_fused_bf16(p, q, uq, st));
        return 0;
    }
    if (bf2 && !x0 && c->fuse_block && c->compute_mode == LASS_COMPUTE_BF16 && cat_in && mh && rb.cin != rb.cout &&
        lass_dec6_fused_bf16_supported(p, q)) {
        ProfScope ps(c, st, P_CONV3X3);
        HIP_TRY(c, lass_launch_dec6_fused_bf16(p, q, st));
        return 0;
    }
    if (int r1 = launch_conv1()) return r1;
    ProfScope ps(c, st, P_CONV3X3);
    if (rb.cin == rb.cout) {
        if (bf2)
            HIP_TRY(c, lass_launch_conv_bf16(x0 ? CONV2_IDENT_PRE : CONV2_IDENT, q, st));
        else if (wino2 && x0 && rb.u2f && lass_wino4_supported(CONV2_IDENT_PRE, q))
            HIP_TRY(c, lass_launch_wino4(CONV2_IDENT_PRE, q, st));
        else if (wino2 && c->wino32 && lass_wino32_supported(x0 ? CONV2_IDENT_PRE : CONV2_IDENT, q))
            HIP_TRY(c, lass_launch_wino32(x0 ? CONV2_IDENT_PRE : CONV2_IDENT, q, st));
        else if (wino2)
            HIP_TRY(c, lass_launch_wino(x0 ? CONV2_IDENT_PRE : CONV2_IDENT, q, st));
        else
            HIP_TRY(c, lass_launch_conv(x0 ? CONV2_IDENT_PRE : CONV2_IDENT, q, st));
    } else {
        if (bf2)
            HIP_TRY(c, lass_launch_conv_bf16(CONV2_SHORTCUT, q, st));
        else if (wino2 && rb.u2f && lass_wino4_supported(CONV2_SHORTCUT, q))
            HIP_TRY(c, lass_launch_wino4(CONV2_SHORTCUT, q, st));
        else if (wino2 && c->wino32 && lass_wino32_supported(CONV2_SHORTCUT, q))
            HIP_TRY(c, lass_launch_wino32(CONV2_SHORTCUT, q, st));
        else if (wino2)
            HIP_TRY(c, lass_launch_wino(CONV2_SHORTCUT, q, st));
        else
            HIP_TRY(c, lass_launch_conv(CONV2_SHORTCUT, q, st));
    }
    return 0;
}

int run_upconv(lass_ctx* c, int di, const float* x, int B, int h, int w, const float* shift, float* out, long out_bs,
               hipStream_t st, const CatCopies* cb = nullptr, bool x_is_act_bf16 = false) {
    const DecSpec& d = c->D[di];
    const Site& s = c->sites[c->dec_site[di]];
    ConvArgs p;
    p.in = x; p.in_bs = (long)d.cin * h * w; p.Cin = d.cin;
    p.w = rawp(c, std::string("base.") + d.name + ".conv1.weight");  // (cin, cout, uh, uw) == [cin][n], n=(co,a,bb)
    p.N = p.Nw = d.cout * d.uh * d.uw;
    p.pro_scale = c->bn_scale + s.off; p.pro_shift = shift + s.off; p.pro_shift_bs = c->n_shift;
    p.out = out; p.out_bs = out_bs; p.B = B; p.H = h; p.W = w; p.up_h = d.uh;
    p.w_bf16 = c->up16[di]; p.w_bf16_lo = c->up16l[di];
    if (x_is_act_bf16) {  // the producer already applied this conv's prologue and wrote blocked bf16
        if (c->compute_mode != LASS_COMPUTE_BF16 || !p.w_bf16 || !lass_bf16_supported(p))
            return fail(c, LASS_ERR_STATE, "activated bf16 input needs the bf16 kernels");
        p.in_bf16 = x;
    }
    if (cb) {  // concat channels [0, C) as the two blocked copies instead of f32
        if (c->compute_mode != LASS_COMPUTE_BF16 || !p.w_bf16 || !lass_bf16_supported(p))
            return fail(c, LASS_ERR_STATE, "blocked bf16 concat copies need the bf16 kernels");
        p.out = nullptr;
        p.out_bf16 = cb->raw; p.out_bf16_act = cb->act; p.out_oct0 = 0; p.out_noct = cb->noct;
        p.act_scale = cb->scale; p.act_shift = cb->shift; p.act_shift_bs = c->n_shift;
    }
    ProfScope ps(c, st, P_TCONV);
    if (c->compute_mode != LASS_COMPUTE_F32 && p.w_bf16 && lass_bf16_supported(p))
        HIP_TRY(c, lass_launch_conv_bf16(TCONV_ACT, p, st));
    else
        HIP_TRY(c, lass_launch_conv(TCONV_ACT, p, st));
    return 0;
}

// ---- workspace plan ---------------------------------------------------------------------------------------------
struct Plan {
    int B, L, T, Tp;
    size_t total = 0;
    size_t mag, cosv, sinv, x0[kMaxBranches], shift, xpre, a2, cat[6], pool[6], center, decout[6], oreal, oimag;
    int eh[7], ew[7];  // encoder block spatial sizes
};

size_t bump(size_t& total, size_t floats) {
    const size_t off = total;
    total += (floats * sizeof(float) + 255) / 256 * 256;
    return off;
}

// The conv kernels address one clip's tensors through 32-bit buffer descriptors and byte offsets, so the largest per-clip
// tensor (decoder_block6's concat at full resolution, f32) bounds the clip length: below 4 GiB for the f32 Winograd
// and the bf16 kernels (all offset arithmetic unsigned; both exercised by the 3.15-GB concat of the 30 s @ 32 kHz multi-STFT
// clip), below 2 GiB for the direct f32 kernels (LASS_WINO=0).  ResUNet30 at 16 kHz: 131 072 B per frame -> 2^32 at 32 768
// frames (327 s); the multi-STFT model (128 ch x 1024 bins): 524 288 B per frame ->
// 2^32 at 8 192 frames (40.9 s at 32 kHz).  Longer inputs go through chunk_inference.
int make_plan(const lass_ctx* c, int B, int L, Plan* pl) {
    const Geometry& g = c->g;
    if (B <= 0 || L <= g.nfft / 2) return LASS_ERR_ARG;
    pl->B = B; pl->L = L;
    pl->T = 1 + L / LASS_HOP;
    pl->Tp = (pl->T + 31) / 32 * 32;
    const size_t clip_max = (size_t)c->dec_cat[5] * pl->Tp * g.fcrop * sizeof(float);
    const size_t limit = (c->compute_mode != LASS_COMPUTE_F32 || c->wino) ? 0xFFFF0000ull : 0x7FFFFFFFull;
    if (clip_max > limit) return LASS_ERR_ARG;
    size_t& t = pl->total;
    t = 0;
    const size_t spec = (size_t)B * pl->T * g.nbins;
    pl->mag = bump(t, spec); pl->cosv = bump(t, spec); pl->sinv = bump(t, spec);
    for (int k = 0; k < g.nbr; ++k) pl->x0[k] = bump(t, (size_t)B * pl->Tp * g.fcrop);
    pl->shift = bump(t, (size_t)B * c->n_shift);
    pl->xpre = bump(t, c->fuse_preconv ? 64 : (size_t)B * kPreCh * pl->Tp * g.fcrop);
    int h = pl->Tp, w = g.fcrop;
    size_t a2max = 0;
    for (int i = 0; i < 7; ++i) {
        pl->eh[i] = h; pl->ew[i] = w;
        const size_t o = (size_t)B * c->E[i].cout * h * w;
        if (o > a2max) a2max = o;
        h /= c->E[i].dh; w /= c->E[i].dw;
        if (i < 6) pl->pool[i] = bump(t, (size_t)B * c->E[i].cout * (i == 0 ? g.nbr : 1) * h * w);
    }
    pl->center = bump(t, (size_t)B * c->E[6].cout * pl->eh[6] * pl->ew[6]);
    for (int d = 0; d < 6; ++d) {
        const int e = 5 - d;  // decoder d concatenates the skip of encoder e
        const size_t hw = (size_t)pl->eh[e] * pl->ew[e];
        pl->cat[d] = bump(t, (size_t)B * c->dec_cat[d] * hw);
        pl->decout[d] = bump(t, (size_t)B * c->D[d].cout * hw);
    }
    pl->a2 = bump(t, a2max);
    pl->oreal = bump(t, spec); pl->oimag = bump(t, spec);
    return 0;
}

// Every entry that launches or allocates runs on the context's device, whatever the caller's current device is.
int use_device(lass_ctx* c) {
    if (!c) return LASS_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    return 0;
}

int check_ready(lass_ctx* c) {
    if (!c) return LASS_ERR_ARG;
    if (!c->finalized) return fail(c, LASS_ERR_STATE, "lass_finalize has not been called (or a parameter changed since)");
    return use_device(c);
}

// Destroys every instantiated graph (live and retired).  Caller has synchronised the device.
void drop_graphs(lass_ctx* c) {
    for (auto& e : c->g_slots) {
        if (e.exec) (void)hipGraphExecDestroy(e.exec);
        e = lass_ctx::GraphEntry();
    }
    for (hipGraphExec_t x : c->g_retired) (void)hipGraphExecDestroy(x);
    c->g_retired.clear();
}

// A batch is split into two overlapping half-batches when it is large enough for each half to fill the GPU on its own
bool split_halves(const lass_ctx* c, int B) { return c->split_batch > 0 && !c->profiling && B >= 8 && (B % 2) == 0; }
int split_parts(const lass_ctx* c, int B) {
    if (!split_halves(c, B)) return 1;
    return c->split_parts == 4 && B >= 16 && (B % 4) == 0 ? 4 : 2;
}

const ResBlock* find_block(const lass_ctx* c, const std::string& prefix) {
    for (const auto& rb : c->enc) if (rb.prefix == prefix) return &rb;
    for (const auto& rb : c->dec) if (rb.prefix == prefix) return &rb;
    return nullptr;
}

}  // namespace

extern "C" {

int lass_version(void) { return 10200; }  // 1.2.0: F(4x4,3x3) kernels, lass_set_graph_replay (1.1.0: multi-STFT model, fused iSTFT, graph replay)

const char* lass_last_error(const lass_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

static int create_impl(lass_ctx** out, int device_id, const Geometry& geom) {
    if (!out) return LASS_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("no HIP device available: ") + hipGetErrorString(e) +
                       " (liblass_hip has no CPU fallback)";
        return LASS_ERR_HIP;
    }
    if (device_id < 0 || device_id >= ndev) {
        g_create_err = "device_id out of range";
        return LASS_ERR_ARG;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) {
        g_create_err = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return LASS_ERR_HIP;
    }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        g_create_err = std::string("device is ") + prop.gcnArchName + "; liblass_hip is built for gfx950 only";
        return LASS_ERR_HIP;
    }
    lass_ctx* c = new lass_ctx();
    c->device = device_id;
    c->g = geom;
    if (const char* e = getenv("LASS_WINO")) c->wino = atoi(e) != 0;
    if (const char* e = getenv("LASS_WINO32")) c->wino32 = atoi(e) != 0;
    if (const char* e = getenv("LASS_FUSE_POOL")) c->fuse_pool = atoi(e) != 0;
    if (const char* e = getenv("LASS_FUSE_MASK")) c->fuse_mask = atoi(e) != 0;
    if (const char* e = getenv("LASS_FUSE_CATB")) c->fuse_catb = atoi(e) != 0;
    if (const char* e = getenv("LASS_FUSE_BLOCK")) c->fuse_block = atoi(e) != 0;
    if (const char* e = getenv("LASS_FUSE_UP")) c->fuse_up = atoi(e) != 0;
    if (const char* e = getenv("LASS_WINO4")) c->wino4_mincin = atoi(e);
    if (const char* e = getenv("LASS_SPLIT")) c->split_batch = std::max(0, std::min(2, atoi(e)));
    if (const char* e = getenv("LASS_SPLIT_PARTS")) c->split_parts = atoi(e) == 4 ? 4 : 2;
    if (const char* e = getenv("LASS_FUSE_PRECONV")) c->fuse_preconv = atoi(e) != 0;
    if (const char* e = getenv("LASS_GRAPH")) c->use_graph = atoi(e) != 0;
    c->prof.resize(P_COUNT);
    for (int i = 0; i < P_COUNT; ++i) c->prof[i].name = kProfNames[i];
    build_arch(c);
    if (hipSetDevice(device_id) != hipSuccess) {
        g_create_err = "hipSetDevice failed";
        delete c;
        return LASS_ERR_HIP;
    }
    // FFT twiddles (the periodic Hann windows are read off the same table), evaluated in double on the host.
    std::vector<float2> tw2k(2048);
    for (int k = 0; k < 2048; ++k) {
        const double a = 2.0 * M_PI * k / 2048.0;
        tw2k[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    if (hipMalloc((void**)&c->tw2k, sizeof(float2) * 2048) != hipSuccess ||
        hipMemcpy(c->tw2k, tw2k.data(), sizeof(float2) * 2048, hipMemcpyHostToDevice) != hipSuccess) {
        g_create_err = "allocating FFT tables failed";
        delete c;
        return LASS_ERR_HIP;
    }
    *out = c;
    return 0;
}

int lass_create(lass_ctx** out, int device_id) { return create_impl(out, device_id, Geometry()); }

int lass_create_multistft(lass_ctx** out, int device_id, int n_fft, int n_windows, const int* win_lengths,
                          int mask_window) {
    if (!out) return LASS_ERR_ARG;
    *out = nullptr;
    Geometry g;
    g.variant = 1;
    g.magphase_sem = 1;
    if (n_fft != 2048 || n_windows < 1 || n_windows > kMaxBranches || !win_lengths) {
        g_create_err = "lass_create_multistft: n_fft must be 2048 and 1 <= n_windows <= 4";
        return LASS_ERR_ARG;
    }
    g.nfft = n_fft; g.nbins = n_fft / 2 + 1; g.fcrop = n_fft / 2; g.nbr = n_windows; g.mask_br = -1;
    for (int k = 0; k < n_windows; ++k) {
        const int w = win_lengths[k];
        if (w < 32 || w > n_fft || (2048 % w) != 0) {
            g_create_err = "lass_create_multistft: window lengths must be powers of two in [32, n_fft]";
            return LASS_ERR_ARG;
        }
        for (int j = 0; j < k; ++j)
            if (g.wins[j] == w) {
                g_create_err = "lass_create_multistft: duplicate window length";
                return LASS_ERR_ARG;
            }
        g.wins[k] = w;
        if (w == mask_window) g.mask_br = k;
    }
    if (g.mask_br < 0) {
        g_create_err = "lass_create_multistft: mask_window is not one of the analysis windows";
        return LASS_ERR_ARG;
    }
    return create_impl(out, device_id, g);
}

// Process-wide guard for the one kernel file that carries packed-f32 arithmetic (wino32.hip's hand-written float2 transform).
// On gfx950 a wave executing v_pk_*_f32 beside a workgroup that feeds v_mfma_f32_32x32x16_bf16 from LDS returned wrong values
// (the x half of a packed result, lanes 48-63, from right operands).  Every other kernel is built without
// packed f32 (-fno-slp-vectorize + ISA audit); wino32.hip runs only in f32 contexts behind LASS_WINO4 != 32.  Two contexts of one
// process may launch on two streams, so a context that can route to wino32.hip and a context with bf16 MFMA kernels must not be
// alive together: the second one to finalize is refused.
static std::atomic<int> g_live_bf16{0}, g_live_w32{0};
static void packed_guard_leave(lass_ctx* c) {
    if (c->live_class == 1) --g_live_bf16;
    if (c->live_class == 2) --g_live_w32;
    c->live_class = 0;
}
static bool routes_to_wino32(const lass_ctx* c, int compute_mode) {
    // wino4.hip takes every layer wino32.hip could serve when its threshold is at most the 32 channels of those layers
    return compute_mode == LASS_COMPUTE_F32 && c->wino && c->wino32 && (c->wino4_mincin <= 0 || c->wino4_mincin > kPreCh);
}
static int packed_guard_enter(lass_ctx* c, int compute_mode) {
    packed_guard_leave(c);
    if (compute_mode != LASS_COMPUTE_F32) {
        if (g_live_w32.load() > 0)
            return fail(c, LASS_ERR_STATE, "lass_finalize: a context of this process routes launches to wino32.hip (LASS_WINO4), whose packed-f32 "
                                           "arithmetic is unsafe beside bf16 MFMA kernels on gfx950 - destroy it first, or leave LASS_WINO4 at its default");
        ++g_live_bf16;
        c->live_class = 1;
    } else if (routes_to_wino32(c, compute_mode)) {
        if (g_live_bf16.load() > 0)
            return fail(c, LASS_ERR_STATE, "lass_finalize: LASS_WINO4 routes this f32 context to wino32.hip, whose packed-f32 arithmetic is unsafe "
                                           "beside the bf16 MFMA kernels of another live context on gfx950 - destroy that context first, or set LASS_WINO32=0");
        ++g_live_w32;
        c->live_class = 2;
    }
    return 0;
}

int lass_destroy(lass_ctx* c) {
    if (!c) return LASS_ERR_ARG;
    packed_guard_leave(c);
    (void)hipSetDevice(c->device);
    free_owned(c);
    for (auto& kv : c->raw) (void)hipFree(kv.second.d);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    (void)hipDeviceSynchronize();  // no replay of a graph below is in flight any more
    drop_graphs(c);
    if (c->g_stream) (void)hipStreamDestroy(c->g_stream);
    for (int i = 0; i < 3; ++i) {
        if (c->s2[i]) (void)hipStreamDestroy(c->s2[i]);
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    (void)hipFree(c->tw2k);
    delete c;
    return 0;
}

int lass_set_param(lass_ctx* c, const char* name_c, const void* data, const int64_t* shape, int ndim, int dtype) {
    if (!c || !name_c || !data || ndim < 0 || (ndim > 0 && !shape)) return fail(c, LASS_ERR_ARG, "bad argument");
    const std::string name(name_c);
    std::vector<int64_t> want = expected_shape(c, name);
    if (want.empty()) {
        if (ignorable(name)) return 1;
        return fail(c, LASS_ERR_ARG, "unknown parameter '" + name + "'");
    }
    if (dtype != LASS_F32) return fail(c, LASS_ERR_ARG, "parameter '" + name + "' must be float32");
    std::vector<int64_t> got(shape, shape + ndim);
    if (got != want) {
        std::string m = "shape mismatch for '" + name + "': got (";
        for (auto v : got) m += std::to_string(v) + ",";
        m += ") want (";
        for (auto v : want) m += std::to_string(v) + ",";
        return fail(c, LASS_ERR_ARG, m + ")");
    }
    size_t n = 1;
    for (auto v : want) n *= (size_t)v;
    HIP_TRY(c, hipSetDevice(c->device));
    Raw& r = c->raw[name];
    if (!r.d) HIP_TRY(c, hipMalloc((void**)&r.d, n * sizeof(float)));
    r.shape = want;
    r.n = n;
    HIP_TRY(c, hipMemcpy(r.d, data, n * sizeof(float), hipMemcpyDefault));
    c->finalized = false;
    return 0;
}

int lass_finalize(lass_ctx* c, int compute_mode) {
    if (!c) return LASS_ERR_ARG;
    if (compute_mode != LASS_COMPUTE_F32 && compute_mode != LASS_COMPUTE_BF16 && compute_mode != LASS_COMPUTE_BF16X3)
        return fail(c, LASS_ERR_ARG, "unsupported compute mode");
    if (int r = packed_guard_enter(c, compute_mode)) return r;
    c->last_split.clear();
    c->compute_mode = compute_mode;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());  // replays of graphs that hold the old derived buffers have drained
    drop_graphs(c);
    free_owned(c);
    c->finalized = false;
    hipStream_t st = nullptr;
    auto need = [&](const std::string& n) -> const float* {
        const float* p = rawp(c, n);
        if (!p && c->err.empty()) c->err = "missing parameter '" + n + "'";
        return p;
    };
    c->err.clear();
    // -- BN folding for the 32 live sites + bn0, FiLM concatenation
    if (dev_alloc(c, &c->bn_scale, c->n_shift) || dev_alloc(c, &c->bn_base, c->n_shift) ||
        dev_alloc(c, &c->film_W, (size_t)c->n_shift * LASS_COND) || dev_alloc(c, &c->film_b, c->n_shift) ||
        dev_alloc(c, &c->bn0_s, c->g.nbins) || dev_alloc(c, &c->bn0_h, c->g.nbins))
        return LASS_ERR_HIP;
    for (const auto& s : c->sites) {
        const float *g = need(s.bn + ".weight"), *b = need(s.bn + ".bias"), *m = need(s.bn + ".running_mean"),
                    *v = need(s.bn + ".running_var"), *fw = need("film." + s.film + ".weight"),
                    *fb = need("film." + s.film + ".bias");
        if (!g || !b || !m || !v || !fw || !fb) return LASS_ERR_STATE;
        HIP_TRY(c, lass_launch_bnfold(g, b, m, v, s.C, kBnEps, c->bn_scale + s.off, c->bn_base + s.off, st));
        HIP_TRY(c, hipMemcpyAsync(c->film_W + (size_t)s.off * LASS_COND, fw, (size_t)s.C * LASS_COND * sizeof(float),
                                  hipMemcpyDeviceToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->film_b + s.off, fb, s.C * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    {
        const float *g = need("base.bn0.weight"), *b = need("base.bn0.bias"), *m = need("base.bn0.running_mean"),
                    *v = need("base.bn0.running_var");
        if (!g || !b || !m || !v) return LASS_ERR_STATE;
        HIP_TRY(c, lass_launch_bnfold(g, b, m, v, c->g.nbins, kBnEps, c->bn0_s, c->bn0_h, st));
    }
    for (int k = 0; k < c->g.nbr; ++k)
        if (!need(c->pre_name[k] + ".weight") || !need(c->pre_name[k] + ".bias")) return LASS_ERR_STATE;
    if (!need("base.after_conv.weight") || !need("base.after_conv.bias")) return LASS_ERR_STATE;
    // -- conv weights -> [cin][tap][cout]
    auto prep = [&](ResBlock& rb) -> int {
        const float *w1 = need(rb.prefix + ".conv1.weight"), *w2 = need(rb.prefix + ".conv2.weight");
        if (!w1 || !w2) return LASS_ERR_STATE;
        if (dev_alloc(c, &rb.w1, (size_t)rb.cout * rb.cin * 9) || dev_alloc(c, &rb.w2, (size_t)rb.cout * rb.cout * 9))
            return LASS_ERR_HIP;
        HIP_TRY(c, lass_launch_relayout_conv(w1, rb.cout, rb.cin, 9, rb.w1, st));
        HIP_TRY(c, lass_launch_relayout_conv(w2, rb.cout, rb.cout, 9, rb.w2, st));
        rb.u1 = rb.u2 = rb.usc = nullptr;
        rb.u1r = rb.u2r = rb.uscr = nullptr;
        rb.u1f = rb.u2f = nullptr;
        rb.b1 = rb.b2 = rb.bsc16 = rb.b1l = rb.b2l = rb.bscl = nullptr;
        const bool bfm = c->compute_mode == LASS_COMPUTE_BF16 || c->compute_mode == LASS_COMPUTE_BF16X3;
        const bool split = c->compute_mode == LASS_COMPUTE_BF16X3;
        if (bfm && rb.cin % 16 == 0 && rb.cout % 16 == 0) {
            unsigned short *t1 = nullptr, *t2 = nullptr;
            if (dev_alloc(c, &t1, (size_t)rb.cout * rb.cin * 9) || dev_alloc(c, &t2, (size_t)rb.cout * rb.cout * 9))
                return LASS_ERR_HIP;
            HIP_TRY(c, lass_launch_weights_bf16(w1, rb.cout, rb.cin, 9, t1, 0, 0, st));
            HIP_TRY(c, lass_launch_weights_bf16(w2, rb.cout, rb.cout, 9, t2, 0, 0, st));
            rb.b1 = t1; rb.b2 = t2;
            if (split) {
                unsigned short *l1 = nullptr, *l2 = nullptr;
                if (dev_alloc(c, &l1, (size_t)rb.cout * rb.cin * 9) || dev_alloc(c, &l2, (size_t)rb.cout * rb.cout * 9))
                    return LASS_ERR_HIP;
                HIP_TRY(c, lass_launch_weights_bf16(w1, rb.cout, rb.cin, 9, l1, 1, 0, st));
                HIP_TRY(c, lass_launch_weights_bf16(w2, rb.cout, rb.cout, 9, l2, 1, 0, st));
                rb.b1l = l1; rb.b2l = l2;
            }
        }
        if (c->wino && c->compute_mode == LASS_COMPUTE_F32) {
            if (dev_alloc(c, &rb.u1, (size_t)16 * rb.cout * rb.cin) || dev_alloc(c, &rb.u2, (size_t)16 * rb.cout * rb.cout))
                return LASS_ERR_HIP;
            HIP_TRY(c, lass_launch_wino_weights(w1, rb.cout, rb.cin, rb.u1, st));
            HIP_TRY(c, lass_launch_wino_weights(w2, rb.cout, rb.cout, rb.u2, st));
            // (F(4x4,3x3) tiles need 32-bin multiples - lass_wino4_supported - so the 16-/8-bin levels get no images: 106 MB saved)
            const bool w4_level = rb.width % 32 == 0;
            if (w4_level && c->wino4_mincin > 0 && rb.cin >= c->wino4_mincin && rb.cin % 8 == 0 && rb.cout % 32 == 0) {
                if (dev_alloc(c, &rb.u1f, (size_t)36 * rb.cout * rb.cin)) return LASS_ERR_HIP;
                HIP_TRY(c, lass_launch_wino4_weights(w1, rb.cout, rb.cin, rb.u1f, st));
            }
            // conv2: the blocks with a 1x1 shortcut, and encoder_block1 (32 -> 32, residual = pre_conv(x0)); the identity blocks
            // at the bottom of the U-Net (16 / 8 bins) stay with wino.hip
            if (w4_level && c->wino4_mincin > 0 && rb.cout >= c->wino4_mincin && (rb.cin != rb.cout || rb.cout == kPreCh) && rb.cout % 32 == 0 &&
                rb.cin % 8 == 0) {
                if (dev_alloc(c, &rb.u2f, (size_t)36 * rb.cout * rb.cout)) return LASS_ERR_HIP;
                HIP_TRY(c, lass_launch_wino4_weights(w2, rb.cout, rb.cout, rb.u2f, st));
            }
            // full-resolution 32-channel blocks, and encoder_block2 (32 -> 64) as two 32-cout slices: wino32.hip
            if ((rb.cout == 32 && rb.cin % 8 == 0) || (rb.cout == 64 && rb.cin == 32)) {
                const size_t ns = rb.cout / 32;
                if (dev_alloc(c, &rb.u1r, ns * 512 * rb.cin) || dev_alloc(c, &rb.u2r, ns * 512 * rb.cout)) return LASS_ERR_HIP;
                HIP_TRY(c, lass_launch_wino32_weights(w1, rb.cout, rb.cin, rb.u1r, st));
                HIP_TRY(c, lass_launch_wino32_weights(w2, rb.cout, rb.cout, rb.u2r, st));
            } else if (rb.cout == 128 && rb.cin == 64) {  // encoder_block3.conv1 as four slices (conv2's weights do not fit)
                if (dev_alloc(c, &rb.u1r, (size_t)4 * 512 * rb.cin)) return LASS_ERR_HIP;
                HIP_TRY(c, lass_launch_wino32_weights(w1, rb.cout, rb.cin, rb.u1r, st));
            }
        }
        rb.wsc = nullptr;
        rb.bsc = nullptr;
        if (rb.cin != rb.cout) {
            const float *ws = need(rb.prefix + ".shortcut.weight"), *bs = need(rb.prefix + ".shortcut.bias");
            if (!ws || !bs) return LASS_ERR_STATE;
            if (dev_alloc(c, &rb.wsc, (size_t)rb.cout * rb.cin)) return LASS_ERR_HIP;
            HIP_TRY(c, lass_launch_relayout_conv(ws, rb.cout, rb.cin, 1, rb.wsc, st));
            rb.bsc = bs;
            if (bfm && rb.b1) {
                unsigned short* t3 = nullptr;
                if (dev_alloc(c, &t3, (size_t)rb.cout * rb.cin)) return LASS_ERR_HIP;
                HIP_TRY(c, lass_launch_weights_bf16(ws, rb.cout, rb.cin, 1, t3, 0, 0, st));
                rb.bsc16 = t3;
                if (split) {
                    unsigned short* l3 = nullptr;
                    if (dev_alloc(c, &l3, (size_t)rb.cout * rb.cin)) return LASS_ERR_HIP;
                    HIP_TRY(c, lass_launch_weights_bf16(ws, rb.cout, rb.cin, 1, l3, 1, 0, st));
                    rb.bscl = l3;
                }
            }
            if (c->wino && c->compute_mode == LASS_COMPUTE_F32) {
                if (dev_alloc(c, &rb.usc, (size_t)4 * rb.cout * rb.cin)) return LASS_ERR_HIP;
                HIP_TRY(c, lass_launch_wino_shortcut_weights(ws, rb.cout, rb.cin, rb.usc, st));
                if ((rb.cout == 32 && rb.cin % 8 == 0) || (rb.cout == 64 && rb.cin == 32)) {
                    if (dev_alloc(c, &rb.uscr, (size_t)(rb.cout / 32) * 128 * rb.cin)) return LASS_ERR_HIP;
                    HIP_TRY(c, lass_launch_wino32_shortcut_weights(ws, rb.cout, rb.cin, rb.uscr, st));
                }
            }
        }
        return 0;
    };
    for (auto& rb : c->enc) { int r = prep(rb); if (r) return r; }
    for (auto& rb : c->dec) { int r = prep(rb); if (r) return r; }
    for (int i = 0; i < 6; ++i) {
        const auto& d = c->D[i];
        const float* wu = need(std::string("base.") + d.name + ".conv1.weight");
        if (!wu) return LASS_ERR_STATE;
        c->up16[i] = c->up16l[i] = nullptr;
        if (c->compute_mode != LASS_COMPUTE_F32 && d.cin % 16 == 0) {
            const int N = d.cout * d.uh * d.uw;
            unsigned short *t = nullptr, *l = nullptr;
            if (dev_alloc(c, &t, (size_t)N * d.cin)) return LASS_ERR_HIP;
            HIP_TRY(c, lass_launch_weights_bf16(wu, N, d.cin, 1, t, 0, 1, st));
            c->up16[i] = t;
            if (c->compute_mode == LASS_COMPUTE_BF16X3) {
                if (dev_alloc(c, &l, (size_t)N * d.cin)) return LASS_ERR_HIP;
                HIP_TRY(c, lass_launch_weights_bf16(wu, N, d.cin, 1, l, 1, 1, st));
                c->up16l[i] = l;
            }
        }
    }
    // decoder_block6's shortcut over the up-sampled half of its concat, composed with the transposed conv (dec6u_fused_bf16_kernel)
    c->up_sc16 = nullptr;
    {
        const DecSpec& d = c->D[5];
        const ResBlock& rb = c->dec[5];
        if (c->compute_mode == LASS_COMPUTE_BF16 && c->g.variant == 0 && d.cin == 64 && d.cout == 32 && d.uh == 2 && d.uw == 2 &&
            rb.cin == 64 && rb.cout == 32 && c->up16[5]) {
            const float* wu = need(std::string("base.") + d.name + ".conv1.weight");
            const float* ws = need(rb.prefix + ".shortcut.weight");
            if (!wu || !ws) return LASS_ERR_STATE;
            float* tmp = nullptr;
            unsigned short* t16 = nullptr;
            if (dev_alloc(c, &tmp, (size_t)4 * rb.cout * d.cin) || dev_alloc(c, &t16, (size_t)4 * rb.cout * d.cin)) return LASS_ERR_HIP;
            HIP_TRY(c, lass_launch_compose_up_shortcut(ws, wu, d.cin, d.cout, rb.cin, rb.cout, tmp, st));
            HIP_TRY(c, lass_launch_weights_bf16(tmp, 4 * rb.cout, d.cin, 1, t16, 0, 0, st));
            c->up_sc16 = t16;
        }
    }
    HIP_TRY(c, hipStreamSynchronize(st));
    c->finalized = true;
    ++c->gen;  // any captured graph refers to the previous derived buffers
    return 0;
}

int lass_workspace_bytes(const lass_ctx* c, int B, int L, size_t* bytes) {
    if (!c || !bytes) return LASS_ERR_ARG;
    Plan pl;
    if (make_plan(c, B, L, &pl)) return LASS_ERR_ARG;  // B < 1, L <= 512 or L beyond the 32-bit per-clip addressing limit
    *bytes = pl.total;
    if (const int P = split_parts(c, B); P > 1) {  // the part-batch plans side by side (they differ from the whole plan by alignment padding only)
        Plan ph;
        if (make_plan(c, B / P, L, &ph)) return LASS_ERR_ARG;
        const size_t all = P * ((ph.total + 255) / 256 * 256);
        if (all > *bytes) *bytes = all;
    }
    return 0;
}

int lass_film_width(const lass_ctx* c) { return c ? c->n_shift : LASS_ERR_ARG; }

int lass_film_offset(const lass_ctx* c, const char* site) {
    if (!c || !site) return -1;
    auto it = c->site_idx.find(site);
    return it == c->site_idx.end() ? -1 : c->sites[it->second].off;
}

int lass_film(lass_ctx* c, const float* cond, int B, float* shift, void* stream) {
    int r = check_ready(c);
    if (r) return r;
    if (!cond || !shift || B <= 0) return fail(c, LASS_ERR_ARG, "lass_film: bad argument");
    HIP_TRY(c, lass_launch_film(cond, B, c->film_W, c->film_b, c->bn_base, c->n_shift, shift, (hipStream_t)stream));
    return 0;
}

int lass_film_raw(lass_ctx* c, const float* cond, int B, float* film, void* stream) {
    int r = check_ready(c);
    if (r) return r;
    if (!cond || !film || B <= 0) return fail(c, LASS_ERR_ARG, "lass_film_raw: bad argument");
    HIP_TRY(c, lass_launch_film(cond, B, c->film_W, c->film_b, nullptr, c->n_shift, film, (hipStream_t)stream));
    return 0;
}

int lass_stft_magphase(lass_ctx* c, const float* wav, int B, int L, float* mag, float* cos_out, float* sin_out,
                       float* real_out, float* imag_out, void* stream) {
    if (!c || !wav || B <= 0 || L <= LASS_NFFT / 2) return fail(c, LASS_ERR_ARG, "lass_stft_magphase: bad argument");
    if (int r = use_device(c)) return r;
    const int T = 1 + L / LASS_HOP;
    StftBranch br;
    br.wlen = LASS_NFFT; br.mag = mag; br.cosv = cos_out; br.sinv = sin_out; br.real = real_out; br.imag = imag_out;
    HIP_TRY(c, lass_launch_stft2(wav, B, L, LASS_NFFT, LASS_HOP, T, T, 1, &br, 0, nullptr, nullptr, c->tw2k,
                                 (hipStream_t)stream));
    return 0;
}

int lass_stft_components(lass_ctx* c, const float* wav, int B, int L, int n_fft, int hop, int n_windows,
                         const int* win_lengths, float* const* mag, float* const* cos_out, float* const* sin_out,
                         void* stream) {
    if (!c || !wav || !win_lengths || !mag || !cos_out || !sin_out || B <= 0 || hop <= 0 || n_windows <= 0 ||
        n_windows > LASS_MAX_STFT_WINDOWS || (n_fft != 1024 && n_fft != 2048) || L <= n_fft / 2)
        return fail(c, LASS_ERR_ARG, "lass_stft_components: bad argument (n_fft 1024 or 2048, at most 4 windows, L > n_fft/2)");
    StftBranch br[LASS_MAX_STFT_WINDOWS];
    for (int i = 0; i < n_windows; ++i) {
        const int w = win_lengths[i];
        if (w < 32 || w > n_fft || (2048 % w) != 0)
            return fail(c, LASS_ERR_ARG, "lass_stft_components: window lengths must be powers of two in [32, n_fft]");
        if (!mag[i] || !cos_out[i] || !sin_out[i]) return fail(c, LASS_ERR_ARG, "lass_stft_components: null output");
        br[i].wlen = w; br[i].mag = mag[i]; br[i].cosv = cos_out[i]; br[i].sinv = sin_out[i];
    }
    if (int r = use_device(c)) return r;
    const int T = 1 + L / hop;
    HIP_TRY(c, lass_launch_stft2(wav, B, L, n_fft, hop, T, T, n_windows, br, 1, nullptr, nullptr, c->tw2k,
                                 (hipStream_t)stream));
    return 0;
}

int lass_multi_stft(lass_ctx* c, const float* wav, int B, int L, int hop, int n_windows, const int* win_lengths,
                    float* const* mag, float* const* cos_out, float* const* sin_out, void* stream) {
    if (!c || !wav || !win_lengths || !mag || !cos_out || !sin_out || B <= 0 || hop <= 0 || n_windows <= 0 ||
        n_windows > LASS_MAX_STFT_WINDOWS)
        return fail(c, LASS_ERR_ARG, "lass_multi_stft: bad argument");
    for (int i = 0; i < n_windows; ++i) {
        const int N = win_lengths[i];
        if (N != 256 && N != 512 && N != 1024 && N != 2048)
            return fail(c, LASS_ERR_ARG, "lass_multi_stft: window lengths must be 256, 512, 1024 or 2048");
        if (L <= N / 2) return fail(c, LASS_ERR_ARG, "lass_multi_stft: waveform shorter than the reflect padding");
        if (!mag[i] || !cos_out[i] || !sin_out[i]) return fail(c, LASS_ERR_ARG, "lass_multi_stft: null output");
    }
    if (int r = use_device(c)) return r;
    HIP_TRY(c, lass_launch_multi_stft(wav, B, L, hop, n_windows, win_lengths, c->tw2k, mag, cos_out, sin_out,
                                      (hipStream_t)stream));
    return 0;
}

int lass_istft(lass_ctx* c, const float* real, const float* imag, int B, int T, int L, float* wav, float* frames_ws,
               void* stream) {
    (void)frames_ws;  // the fused kernel needs no frame scratch (kept in the signature for ABI stability; may be NULL)
    return lass_istft_nfft(c, real, imag, B, T, L, LASS_NFFT, LASS_NFFT, wav, stream);
}

int lass_istft_nfft(lass_ctx* c, const float* real, const float* imag, int B, int T, int L, int n_fft, int win_length,
                    float* wav, void* stream) {
    if (!c || !real || !imag || !wav || B <= 0 || T <= 0 || L <= 0 || (n_fft != 1024 && n_fft != 2048) ||
        win_length < 32 || win_length > n_fft || (2048 % win_length) != 0 ||
        (long)L + n_fft / 2 > (long)(T - 1) * LASS_HOP + n_fft)
        return fail(c, LASS_ERR_ARG, "lass_istft: bad argument");
    if (int r = use_device(c)) return r;
    HIP_TRY(c, lass_launch_istft2(real, imag, B, T, L, n_fft, win_length, LASS_HOP, c->tw2k, wav, (hipStream_t)stream));
    return 0;
}

int lass_convblock(lass_ctx* c, const char* prefix, const float* x, int B, int H, int W, const float* shift, float* y,
                   float* scratch, void* stream) {
    int r = check_ready(c);
    if (r) return r;
    if (!prefix || !x || !shift || !y || !scratch) return fail(c, LASS_ERR_ARG, "lass_convblock: bad argument");
    const ResBlock* rb = find_block(c, prefix);
    if (!rb) return fail(c, LASS_ERR_ARG, std::string("lass_convblock: unknown block '") + prefix + "'");
    const long HW = (long)H * W;
    return run_resblock(c, *rb, x, rb->cin * HW, B, H, W, shift, scratch, y, rb->cout * HW, (hipStream_t)stream);
}

int lass_encoder_block(lass_ctx* c, const char* name, const float* x, int B, int H, int W, const float* shift, float* y,
                       float* pool, float* scratch, void* stream) {
    int r = check_ready(c);
    if (r) return r;
    if (!name || !x || !shift || !y || !scratch || B <= 0 || H <= 0 || W <= 0)
        return fail(c, LASS_ERR_ARG, "lass_encoder_block: bad argument");
    for (size_t bi = 0; bi < c->enc.size(); ++bi) {
        const ResBlock& rb = c->enc[bi];
        if (rb.prefix != std::string(name) + ".conv_block1") continue;
        const EncSpec& e = c->E[bi < (size_t)c->g.nbr ? 0 : bi - c->g.nbr + 1];
        const long HW = (long)H * W;
        const bool pooled = e.dw == 2;
        if (pooled && (!pool || W % 2 != 0)) return fail(c, LASS_ERR_ARG, "lass_encoder_block: pool output needed, W even");
        // same rule as lass_separate: the pool rides in conv2's epilogue when the rows divide, else its own kernel
        const bool fuse = pooled && c->fuse_pool && (H % e.dh) == 0;
        hipStream_t st = (hipStream_t)stream;
        r = run_resblock(c, rb, x, rb.cin * HW, B, H, W, shift, scratch, y, rb.cout * HW, st, fuse ? pool : nullptr, e.dh);
        if (r) return r;
        if (pooled && !fuse) HIP_TRY(c, lass_launch_pool(y, rb.cout * HW, B, rb.cout, H, W, e.dh, e.dw, pool, st));
        return 0;
    }
    return fail(c, LASS_ERR_ARG, std::string("lass_encoder_block: unknown encoder '") + name + "'");
}

int lass_front_end(lass_ctx* c, const float* wav, int B, int L, float* mag, float* cos_out, float* sin_out, float* x0,
                   void* stream) {
    int r = check_ready(c);
    if (r) return r;
    const Geometry& g = c->g;
    if (!wav || !x0 || B <= 0 || L <= g.nfft / 2) return fail(c, LASS_ERR_ARG, "lass_front_end: bad argument");
    const int T = 1 + L / LASS_HOP, Tp = (T + 31) / 32 * 32;
    StftBranch br[kMaxBranches];
    for (int k = 0; k < g.nbr; ++k) {  // x0 (nbr, B, Tp, fcrop); mag / cos / sin of the mask branch
        br[k].wlen = g.wins[k];
        br[k].x0 = x0 + (size_t)k * B * Tp * g.fcrop;
        if (k == g.mask_br) { br[k].mag = mag; br[k].cosv = cos_out; br[k].sinv = sin_out; }
    }
    HIP_TRY(c, lass_launch_stft2(wav, B, L, g.nfft, LASS_HOP, T, Tp, g.nbr, br, g.magphase_sem, c->bn0_s, c->bn0_h,
                                 c->tw2k, (hipStream_t)stream));
    return 0;
}

int lass_workspace_tensor(const lass_ctx* c, int B, int L, const char* name_c, size_t* offset, int64_t shape[4],
                          int64_t strides[4]) {
    if (!c || !name_c || !offset || !shape || !strides) return LASS_ERR_ARG;
    // a batch whose LAST separation ran as part-batches (B >= 8, LASS_SPLIT: the replayed graph by default) holds their layouts
    // in its workspace, not this one; after an eager, unsplit call - or before any call - the whole-batch layout is what is there
    if (auto it = c->last_split.find({B, L}); it != c->last_split.end() && it->second) return LASS_ERR_STATE;
    Plan pl;
    if (make_plan(c, B, L, &pl)) return LASS_ERR_ARG;
    const Geometry& g = c->g;
    const std::string name(name_c);
    auto put = [&](size_t off, int64_t C, int64_t H, int64_t W, int64_t bs) {
        *offset = off;
        shape[0] = B; shape[1] = C; shape[2] = H; shape[3] = W;
        strides[0] = bs; strides[1] = H * W; strides[2] = W; strides[3] = 1;
        return 0;
    };
    const int64_t spec = (int64_t)pl.T * g.nbins;
    if (name == "mag") return put(pl.mag, 1, pl.T, g.nbins, spec);
    if (name == "cos") return put(pl.cosv, 1, pl.T, g.nbins, spec);
    if (name == "sin") return put(pl.sinv, 1, pl.T, g.nbins, spec);
    if (name == "out_real") return put(pl.oreal, 1, pl.T, g.nbins, spec);
    if (name == "out_imag") return put(pl.oimag, 1, pl.T, g.nbins, spec);
    for (int k = 0; k < g.nbr; ++k)
        if (name == (g.variant == 0 ? std::string("x0") : "x0." + std::to_string(g.wins[k])))
            return put(pl.x0[k], 1, pl.Tp, g.fcrop, (int64_t)pl.Tp * g.fcrop);
    for (int i = 0; i < 7; ++i) {
        const EncSpec& e = c->E[i];
        const int64_t H = pl.eh[i], W = pl.ew[i];
        const int64_t C = (int64_t)e.cout * (i == 0 ? g.nbr : 1);  // encoder_block1: all branches, channel-concatenated
        if (name == e.name) {
            if (i == 6) return put(pl.center, C, H, W, C * H * W);
            const int d = 5 - i;  // the skip lives in place behind the transposed-conv half of decoder d's concat
            return put(pl.cat[d] + (size_t)c->D[d].cout * H * W * sizeof(float), C, H, W, (int64_t)c->dec_cat[d] * H * W);
        }
        if (i < 6 && name == std::string(e.name) + ".pool")
            return put(pl.pool[i], C, H / e.dh, W / e.dw, C * (H / e.dh) * (W / e.dw));
    }
    for (int d = 0; d < 6; ++d) {
        const int64_t H = pl.eh[5 - d], W = pl.ew[5 - d], C = c->D[d].cout;
        if (name == std::string(c->D[d].name) + ".up") return put(pl.cat[d], C, H, W, (int64_t)c->dec_cat[d] * H * W);
        if (name == c->D[d].name) return put(pl.decout[d], C, H, W, C * H * W);
    }
    return LASS_ERR_ARG;
}

int lass_upconv(lass_ctx* c, const char* name, const float* x, int B, int h, int w, const float* shift, float* y,
                void* stream) {
    int r = check_ready(c);
    if (r) return r;
    if (!name || !x || !shift || !y) return fail(c, LASS_ERR_ARG, "lass_upconv: bad argument");
    for (int i = 0; i < 6; ++i)
        if (std::string("base.") + c->D[i].name == name)
            return run_upconv(c, i, x, B, h, w, shift, y, (long)c->D[i].cout * h * c->D[i].uh * w * c->D[i].uw,
                              (hipStream_t)stream);
    return fail(c, LASS_ERR_ARG, std::string("lass_upconv: unknown decoder '") + name + "'");
}

int lass_mask_apply(lass_ctx* c, const float* x12, const float* mag, const float* cos_in, const float* sin_in, int B,
                    int T, int Tpad, float* out_real, float* out_imag, void* stream) {
    int r = check_ready(c);
    if (r) return r;
    if (!x12 || !mag || !cos_in || !sin_in || !out_real || !out_imag)
        return fail(c, LASS_ERR_ARG, "lass_mask_apply: bad argument");
    HIP_TRY(c, lass_launch_mask(x12, rawp(c, "base.after_conv.weight"), rawp(c, "base.after_conv.bias"), mag, cos_in,
                                sin_in, B, T, Tpad, c->g.fcrop, out_real, out_imag, (hipStream_t)stream));
    return 0;
}

int lass_sdr_stats(lass_ctx* c, const float* ref, const float* est, int B, int L, double* stats, void* stream) {
    if (!c || !ref || !est || !stats || B <= 0 || L <= 0) return fail(c, LASS_ERR_ARG, "lass_sdr_stats: bad argument");
    if (int r = use_device(c)) return r;
    HIP_TRY(c, lass_launch_sdr(ref, est, B, L, stats, (hipStream_t)stream));
    return 0;
}

int lass_mix_at_snr(lass_ctx* c, float* source, const float* noise, const float* snr_db, float* mixture, int B, int L,
                    double* scratch, void* stream) {
    if (!c || !source || !noise || !snr_db || !mixture || !scratch || B <= 0 || L <= 0)
        return fail(c, LASS_ERR_ARG, "lass_mix_at_snr: bad argument");
    if (int r = use_device(c)) return r;
    HIP_TRY(c, lass_launch_mix_at_snr(source, noise, snr_db, mixture, B, L, scratch, (hipStream_t)stream));
    return 0;
}

int lass_segment_mix(lass_ctx* c, const float* waveforms, int B, int L, const int* mix_num, const float* comp_db, int max_comp,
                     const float* noise_db, float* mixture, float* segment, double* scratch, void* stream) {
    if (!c || !waveforms || !mix_num || !comp_db || !noise_db || !mixture || !segment || !scratch || B <= 0 || L <= 0)
        return fail(c, LASS_ERR_ARG, "lass_segment_mix: bad argument");
    if (max_comp < 1 || max_comp > 7) return fail(c, LASS_ERR_ARG, "lass_segment_mix: max_comp (max_mix_num - 1) must be 1 ... 7");
    if (mixture == waveforms || segment == waveforms || mixture == segment)
        return fail(c, LASS_ERR_ARG, "lass_segment_mix: waveforms, mixture and segment must be three different buffers");
    if (int r = use_device(c)) return r;
    HIP_TRY(c, lass_launch_segment_mix(waveforms, B, L, mix_num, comp_db, max_comp, noise_db, mixture, segment, scratch,
                                       (hipStream_t)stream));
    return 0;
}

// Precomputed analysis of the mixtures (the reference's multi-STFT wrapper reads these from input_dict,
// resunet_with_multistft.py:233-241): magnitude per branch, cos / sin of the mask branch, each (B, T, nbins).
struct Components {
    const float* mag[kMaxBranches];
    const float* cosv;
    const float* sinv;
};

static int separate_impl(lass_ctx* c, const float* mixture, const Components* comp, const float* condition, float* out,
                         int B, int L, void* workspace, size_t workspace_bytes, void* stream, const char* who) {
    int r = check_ready(c);
    if (r) return r;
    const Geometry& g = c->g;
    if ((!mixture && !comp) || !condition || !out || !workspace) return fail(c, LASS_ERR_ARG, std::string(who) + ": null pointer");
    Plan pl;
    if (make_plan(c, B, L, &pl))
        return fail(c, LASS_ERR_ARG, std::string(who) + ": need B >= 1 and " + std::to_string(g.nfft / 2) +
                                         " < L, with decoder_block6's concat (" + std::to_string(c->dec_cat[5]) +
                                         " ch x frames x " + std::to_string(g.fcrop) + " bins, f32) below " +
                                         ((c->compute_mode != LASS_COMPUTE_F32 || c->wino) ? "4" : "2") +
                                         " GiB per clip (longer clips: ResUNet30.chunk_inference)");
    if (workspace_bytes < pl.total)
        return fail(c, LASS_ERR_WORKSPACE, "workspace too small: need " + std::to_string(pl.total) + " bytes");
    if (((uintptr_t)workspace & 255) != 0) return fail(c, LASS_ERR_ARG, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    auto F = [&](size_t off) { return (float*)(ws + off); };
    const int T = pl.T, Tp = pl.Tp, nbr = g.nbr;
    float* shift = F(pl.shift);
    // ---- front end: STFT + magnitude / phase + bn0 / T-pad / F-crop (base.py:83-113, resunet.py:533-552) ------------
    const float *mag_m = F(pl.mag), *cos_m = F(pl.cosv), *sin_m = F(pl.sinv);  // of the mask branch
    if (mixture) {
        ProfScope ps(c, st, P_STFT);
        StftBranch br[kMaxBranches];
        for (int k = 0; k < nbr; ++k) {
            br[k].wlen = g.wins[k];
            br[k].x0 = F(pl.x0[k]);
            if (k == g.mask_br) { br[k].mag = F(pl.mag); br[k].cosv = F(pl.cosv); br[k].sinv = F(pl.sinv); }
        }
        HIP_TRY(c, lass_launch_stft2(mixture, B, L, g.nfft, LASS_HOP, T, Tp, nbr, br, g.magphase_sem, c->bn0_s, c->bn0_h,
                                     c->tw2k, st));
    } else {
        ProfScope ps(c, st, P_STFT);
        for (int k = 0; k < nbr; ++k) {
            if (!comp->mag[k]) return fail(c, LASS_ERR_ARG, std::string(who) + ": null magnitude");
            HIP_TRY(c, lass_launch_x0_from_mag(comp->mag[k], B, T, Tp, g.fcrop, c->bn0_s, c->bn0_h, F(pl.x0[k]), st));
        }
        if (!comp->cosv || !comp->sinv) return fail(c, LASS_ERR_ARG, std::string(who) + ": null phase");
        mag_m = comp->mag[g.mask_br]; cos_m = comp->cosv; sin_m = comp->sinv;
    }
    {
        ProfScope ps(c, st, P_FILM);
        HIP_TRY(c, lass_launch_film(condition, B, c->film_W, c->film_b, c->bn_base, c->n_shift, shift, st));
    }
    // pre_conv (resunet.py:555) is normally never materialised: encoder_block1 forms it from x0 while staging
    const bool fuse_pre = c->fuse_preconv || nbr > 1;
    // bf16 mode: decoders 2-6 (2x2 up-sampling) take their concat as blocked bf16 copies written by the
    // producers
    CatCopies cb[6];
    bool use_cb[6] = {false, false, false, false, false, false};
    for (int d = 1; d < 6; ++d) {
        const int e = 5 - d;
        const ResBlock& rd = c->dec[d];
        const ResBlock& re = c->enc[nbr - 1 + e];
        const long hw = (long)pl.eh[e] * pl.ew[e];
        use_cb[d] = c->fuse_catb && c->compute_mode == LASS_COMPUTE_BF16 && c->D[d].uh == 2 && c->D[d].uw == 2 &&
                    c->fuse_pool && (pl.eh[e] % c->E[e].dh) == 0 && rd.cout % 16 == 0 && rd.b1 && rd.b2 && rd.bsc16 &&
                    c->up16[d] && re.b1 && re.b2 && (e != 0 || fuse_pre);
        cb[d].act = F(pl.cat[d]);
        cb[d].raw = (char*)F(pl.cat[d]) + (size_t)B * rd.cin * hw * 2;
        cb[d].noct = rd.cin / 8;
        cb[d].scale = c->bn_scale + c->sites[rd.s1].off;
        cb[d].shift = shift + c->sites[rd.s1].off;
    }
    // ... and encoder blocks 2-5 take their (pooled) input the same way, from the previous block's fused pool
    CatCopies pc[4];
    bool use_pc[4] = {false, false, false, false};
    for (int i = 0; i < 4; ++i) {
        const ResBlock& nx = c->enc[nbr - 1 + i + 1];
        const long hwo = (long)pl.eh[i + 1] * pl.ew[i + 1];
        use_pc[i] = c->fuse_catb && c->compute_mode == LASS_COMPUTE_BF16 && c->fuse_pool && c->E[i].dh == 2 &&
                    (pl.eh[i] % 2) == 0 && pl.ew[i] % 32 == 0 && (i != 0 || fuse_pre) && use_cb[5 - i] && use_cb[5 - (i + 1)] &&
                    nx.cin != nx.cout && nx.cin % 16 == 0 && nx.b1 && nx.b2 && nx.bsc16;
        pc[i].act = F(pl.pool[i]);
        pc[i].raw = (char*)F(pl.pool[i]) + (size_t)B * nx.cin * hwo * 2;
        pc[i].noct = nx.cin / 8;
        pc[i].scale = c->bn_scale + c->sites[nx.s1].off;
        pc[i].shift = shift + c->sites[nx.s1].off;
    }
    // ---- encoder (resunet.py:556-562; resunet_with_multistft.py:151-179) -----------------------------------------------
    const float* x = nullptr;
    for (int i = 0; i < 7; ++i) {
        const int H = pl.eh[i], W = pl.ew[i];
        const long HW = (long)H * W;
        const EncSpec& e = c->E[i];
        const int nb = i == 0 ? nbr : 1;  // encoder_block1 runs once per analysis branch
        const bool fuse_pool = i < 6 && c->fuse_pool && (H % e.dh) == 0;
        float* o = nullptr;
        long o_bs = 0;
        for (int k = 0; k < nb; ++k) {
            const ResBlock& rb = c->enc[i == 0 ? k : nbr - 1 + i];
            const int cofs = k * e.cout;  // channel offset of this branch inside the concatenated skip / pool
            if (i < 6) {  // skip output lives behind the transposed-conv half of decoder (5-i)'s concat buffer
                const int d = 5 - i;
                o = F(pl.cat[d]) + (size_t)(c->D[d].cout + cofs) * HW;
                o_bs = (long)c->dec_cat[d] * HW;
            } else {
                o = F(pl.center);
                o_bs = rb.cout * HW;
            }
            const long Ho = H / e.dh, Wo = W / e.dw;
            float* pool_k = i < 6 ? F(pl.pool[i]) + (size_t)cofs * Ho * Wo : nullptr;
            const long pool_bs = (long)e.cout * nb * Ho * Wo;
            PreConv pre{nullptr, nullptr, nullptr};
            const float* xin = x;
            if (i == 0) {
                if (fuse_pre) {
                    pre = PreConv{F(pl.x0[k]), rawp(c, c->pre_name[k] + ".weight"), rawp(c, c->pre_name[k] + ".bias")};
                } else {
                    ProfScope ps(c, st, P_PRECONV);
                    HIP_TRY(c, lass_launch_preconv(F(pl.x0[k]), rawp(c, c->pre_name[k] + ".weight"),
                                                   rawp(c, c->pre_name[k] + ".bias"), B, kPreCh, HW, F(pl.xpre), st));
                    xin = F(pl.xpre);
                }
            }
            // F.avg_pool2d (resunet.py:197) is fused into conv2's epilogue; W >= 16 at every pooled level
            r = run_resblock(c, rb, xin, rb.cin * HW, B, H, W, shift, F(pl.a2), o, o_bs, st, fuse_pool ? pool_k : nullptr,
                             e.dh, pre.x0 ? &pre : nullptr, nullptr, (i < 5 && use_cb[5 - i]) ? &cb[5 - i] : nullptr,
                             (i >= 1 && i <= 4 && use_pc[i - 1]) ? &pc[i - 1] : nullptr, nullptr,
                             (i < 4 && use_pc[i]) ? &pc[i] : nullptr, pool_bs, nullptr, cofs);
            if (r) return r;
            if (i < 6 && !fuse_pool) {
                ProfScope ps(c, st, P_POOL);
                if (nb > 1) return fail(c, LASS_ERR_STATE, "the multi-STFT model needs the fused avg-pool (LASS_FUSE_POOL=1)");
                HIP_TRY(c, lass_launch_pool(o, o_bs, B, rb.cout, H, W, e.dh, e.dw, F(pl.pool[i]), st));
            }
        }
        x = i < 6 ? F(pl.pool[i]) : o;  // conv_block7a: downsample (1,1) is the identity (resunet.py:363-370)
    }
    // ---- decoder (resunet.py:563-568) -------------------------------------------------------------------------
    bool x_act = false;  // x (input of the next transposed conv) is an activated blocked bf16 tensor
    bool fused_head = false;
    for (int d = 0; d < 6; ++d) {
        const int e = 5 - d;
        const int H = pl.eh[e], W = pl.ew[e];
        const long HW = (long)H * W;
        const int h = H / c->D[d].uh, w = W / c->D[d].uw;
        const ResBlock& rb = c->dec[d];
        // decoder_block6 in the blocked bf16 pipeline: the transposed conv runs inside the block's fused kernel, its output
        // (half of the concat, at the full resolution) is never written
        const UpFuse upf{x, c->D[d].cin, h, w, c->up16[d], c->up_sc16};
        const bool fuse_up = d == 5 && c->fuse_up && c->fuse_block && c->fuse_mask && use_cb[d] && x_act && c->up_sc16 &&
                             c->compute_mode == LASS_COMPUTE_BF16 && rb.cout == 32 && rb.cin == 64 && W == g.fcrop && W % 32 == 0 &&
                             H % 2 == 0 && (unsigned long long)H * W * 8ull < 0x10000000ull;
        if (!fuse_up) {
            r = run_upconv(c, d, x, B, h, w, shift, F(pl.cat[d]), rb.cin * HW, st, use_cb[d] ? &cb[d] : nullptr, x_act);
            if (r) return r;
        }
        // this decoder's output feeds only the next transposed conv: hand it over activated, as blocked bf16
        const bool act_next = d < 5 && c->fuse_catb && c->compute_mode == LASS_COMPUTE_BF16 && rb.b1 && rb.b2 &&
                              rb.bsc16 && rb.cout % 16 == 0 && c->up16[d + 1];
        x_act = act_next;
        // decoder_block6 (32 channels at the full resolution): after_conv + mask run in conv2's epilogue
        const MaskHead head{mag_m, cos_m, sin_m, F(pl.oreal), F(pl.oimag), T, g.nbins};
        fused_head = d == 5 && c->fuse_mask && rb.cout == 32 && rb.cin != rb.cout && W == g.fcrop;
        r = run_resblock(c, rb, F(pl.cat[d]), rb.cin * HW, B, H, W, shift, F(pl.a2), F(pl.decout[d]), rb.cout * HW, st,
                         nullptr, 2, nullptr, fused_head ? &head : nullptr, nullptr, use_cb[d] ? &cb[d] : nullptr,
                         act_next ? &c->sites[c->dec_site[d + 1]] : nullptr, nullptr, 0, fuse_up ? &upf : nullptr);
        if (r) return r;
        x = F(pl.decout[d]);
    }
    if (!fused_head) {
        ProfScope ps(c, st, P_MASK);
        HIP_TRY(c, lass_launch_mask(x, rawp(c, "base.after_conv.weight"), rawp(c, "base.after_conv.bias"), mag_m, cos_m,
                                    sin_m, B, T, Tp, g.fcrop, F(pl.oreal), F(pl.oimag), st));
    }
    {
        ProfScope ps(c, st, P_ISTFT);
        HIP_TRY(c, lass_launch_istft2(F(pl.oreal), F(pl.oimag), B, T, L, g.nfft, g.wins[g.mask_br], LASS_HOP, c->tw2k, out, st));
    }
    return 0;
}

// lass_separate's launches, whole or as two overlapping half-batches (lass_ctx::split_batch).  Capture-safe: under stream
// capture the event pair makes `s2` a parallel branch of the same graph.
static int separate_any(lass_ctx* c, const float* mixture, const float* condition, float* out, int B, int L, void* workspace,
                        size_t workspace_bytes, hipStream_t stream, bool capturing) {
    Plan ph;
    const int P = c->finalized ? split_parts(c, B) : 1;
    if (P < 2 || !mixture || !condition || !out || !workspace || (!capturing && c->split_batch < 2) || make_plan(c, B / P, L, &ph) ||
        P * ((ph.total + 255) / 256 * 256) > workspace_bytes) {
        c->last_split[{B, L}] = false;
        return separate_impl(c, mixture, nullptr, condition, out, B, L, workspace, workspace_bytes, stream, "lass_separate");
    }
    c->last_split[{B, L}] = true;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->ev_fork) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < P - 1; ++i)
        if (!c->s2[i]) {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->s2[i], hipStreamNonBlocking));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
        }
    const int h = B / P;
    const size_t part_ws = (ph.total + 255) / 256 * 256;
    HIP_TRY(c, hipEventRecord(c->ev_fork, stream));
    for (int i = 0; i < P - 1; ++i) HIP_TRY(c, hipStreamWaitEvent(c->s2[i], c->ev_fork, 0));
    int r = 0;
    for (int i = 0; i < P; ++i) {
        const int ri = separate_impl(c, mixture + (size_t)i * h * L, nullptr, condition + (size_t)i * h * LASS_COND, out + (size_t)i * h * L,
                                     h, L, (char*)workspace + i * part_ws, part_ws, i ? c->s2[i - 1] : stream, "lass_separate");
        if (!r) r = ri;
    }
    // the joins are recorded even after a failure: a capturing stream must get its branches back
    for (int i = 0; i < P - 1; ++i) {
        HIP_TRY(c, hipEventRecord(c->ev_join[i], c->s2[i]));
        HIP_TRY(c, hipStreamWaitEvent(stream, c->ev_join[i], 0));
    }
    return r;
}

int lass_separate(lass_ctx* c, const float* mixture, const float* condition, float* out, int B, int L, void* workspace,
                  size_t workspace_bytes, void* stream) {
    if (!c) return LASS_ERR_ARG;
    if (!mixture) return fail(c, LASS_ERR_ARG, "lass_separate: null pointer");
    // Graph replay: the third call that presents the same (pointers, shape) key is captured once (on an internal stream; the
    // caller's may be the legacy default stream, which cannot be captured) and replayed from then on.  Callers that hand
    // over fresh buffers every time simply stay on the eager path; so does a profiled context.
    lass_ctx::GraphKey key;
    key.mix = mixture; key.cond = condition; key.out = out; key.ws = workspace; key.B = B; key.L = L; key.gen = c->gen;
#ifdef LASS_CONV_DIAG
    c->use_graph = false;  // the diagnostic launchers allocate and synchronise: not capturable
#endif
    if (c->use_graph && !c->profiling && c->finalized) {
        ++c->g_tick;
        lass_ctx::GraphEntry* slot = nullptr;
        for (auto& e : c->g_slots)
            if (e.seen > 0 && e.key == key) slot = &e;
        if (!slot) {  // take the least recently used slot, preferring one without a graph
            for (auto& e : c->g_slots)
                if (!slot || (!e.exec && slot->exec) || (!e.exec == !slot->exec && e.used < slot->used)) slot = &e;
            if (slot->exec) {  // may still be replaying on some stream: destroyed only behind a device synchronisation
                c->g_retired.push_back(slot->exec);
                if (c->g_retired.size() > 8) {  // a caller cycling through many recurring keys: bound the list here
                    HIP_TRY(c, hipSetDevice(c->device));
                    HIP_TRY(c, hipDeviceSynchronize());
                    for (hipGraphExec_t x : c->g_retired) (void)hipGraphExecDestroy(x);
                    c->g_retired.clear();
                }
            }
            *slot = lass_ctx::GraphEntry();
            slot->key = key;
        }
        slot->used = c->g_tick;
        ++slot->seen;
        if (slot->exec) {
            if (workspace_bytes < slot->need)
                return fail(c, LASS_ERR_WORKSPACE, "workspace too small: need " + std::to_string(slot->need) + " bytes");
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipGraphLaunch(slot->exec, (hipStream_t)stream));
            ++c->g_replays;
            c->last_split[{B, L}] = slot->split;
            return 0;
        }
        if (slot->seen >= 3) {  // worth a capture
            // argument errors are reported as such, before any capture starts: only a failure of the capture machinery
            // itself (begin / end / instantiate, or a launch refused under capture) turns graph replay off
            {
                int r0 = check_ready(c);
                if (r0) return r0;
                Plan pl0;
                if (!condition || !out || !workspace || make_plan(c, B, L, &pl0) || workspace_bytes < pl0.total ||
                    ((uintptr_t)workspace & 255) != 0)
                    return separate_any(c, mixture, condition, out, B, L, workspace, workspace_bytes, (hipStream_t)stream, false);
                slot->need = pl0.total;
                if (const int P = split_parts(c, B); P > 1) {  // (checked again at capture time by separate_any: smaller workspaces run unsplit)
                    Plan ph0;
                    if (!make_plan(c, B / P, L, &ph0) && P * ((ph0.total + 255) / 256 * 256) <= workspace_bytes)
                        slot->need = std::max(slot->need, P * ((ph0.total + 255) / 256 * 256));
                }
            }
            HIP_TRY(c, hipSetDevice(c->device));
            if (!c->g_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->g_stream, hipStreamNonBlocking));
            bool ok = false;
            if (hipStreamBeginCapture(c->g_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int r = separate_any(c, mixture, condition, out, B, L, workspace, workspace_bytes, c->g_stream, true);
                hipGraph_t graph = nullptr;
                const hipError_t e = hipStreamEndCapture(c->g_stream, &graph);  // always ends the capture, also after a failure
                if (r == 0 && e == hipSuccess && graph && hipGraphInstantiate(&slot->exec, graph, nullptr, nullptr, 0) == hipSuccess)
                    ok = true;
                if (auto it = c->last_split.find({B, L}); it != c->last_split.end()) slot->split = it->second;
                else
                    slot->exec = nullptr;
                if (graph) (void)hipGraphDestroy(graph);
            }
            if (ok) {
                ++c->g_captures;
                HIP_TRY(c, hipGraphLaunch(slot->exec, (hipStream_t)stream));
                return 0;
            }
            // A launch failed inside the capture, or the runtime refused it: nothing has run yet.  Stay eager from now on and
            // run this call eagerly below (an error is reported only if the eager run fails too).
            (void)hipGetLastError();
            c->use_graph = false;
            c->err.clear();
        }
    }
    return separate_any(c, mixture, condition, out, B, L, workspace, workspace_bytes, (hipStream_t)stream, false);
}

int lass_set_graph_replay(lass_ctx* c, int enabled) {
    if (!c) return LASS_ERR_ARG;
    c->use_graph = enabled != 0;  // captured graphs stay cached: switching back on replays them again
    return 0;
}

int lass_graph_stats(const lass_ctx* c, long* captures, long* replays) {
    if (!c) return LASS_ERR_ARG;
    if (captures) *captures = c->g_captures;
    if (replays) *replays = c->g_replays;
    return c->use_graph ? 1 : 0;
}

int lass_separate_components(lass_ctx* c, const float* const* mag, const float* cos_mask, const float* sin_mask,
                             const float* condition, float* out, int B, int L, void* workspace, size_t workspace_bytes,
                             void* stream) {
    if (!c) return LASS_ERR_ARG;
    if (!mag || !cos_mask || !sin_mask) return fail(c, LASS_ERR_ARG, "lass_separate_components: null pointer");
    Components comp;
    for (int k = 0; k < kMaxBranches; ++k) comp.mag[k] = k < c->g.nbr ? mag[k] : nullptr;
    comp.cosv = cos_mask; comp.sinv = sin_mask;
    return separate_impl(c, nullptr, &comp, condition, out, B, L, workspace, workspace_bytes, stream,
                         "lass_separate_components");
}

int lass_set_profiling(lass_ctx* c, int enabled) {
    if (!c) return LASS_ERR_ARG;
    if (enabled && c->ev_pool.size() < 2 * kProfPairs) {
        HIP_TRY(c, hipSetDevice(c->device));
        c->pending.reserve(kProfPairs);
        while (c->ev_pool.size() < 2 * kProfPairs) {
            hipEvent_t e;
            HIP_TRY(c, hipEventCreate(&e));
            c->ev_pool.push_back(e);
        }
    }
    c->profiling = enabled != 0;
    return 0;
}
int lass_profile_count(const lass_ctx* c) { return c ? P_COUNT : LASS_ERR_ARG; }
int lass_profile_get(lass_ctx* c, int i, const char** name, double* ms, int* launches) {
    if (!c || i < 0 || i >= P_COUNT) return LASS_ERR_ARG;
    prof_collect(c);
    if (name) *name = c->prof[i].name;
    if (ms) *ms = c->prof[i].ms;
    if (launches) *launches = c->prof[i].launches;
    return 0;
}
int lass_profile_reset(lass_ctx* c) {
    if (!c) return LASS_ERR_ARG;
    prof_collect(c);
    for (auto& p : c->prof) { p.ms = 0; p.launches = 0; }
    return 0;
}

}  // extern "C"
