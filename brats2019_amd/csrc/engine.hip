// engine.hip -- C-ABI entry points of libresunet_hip.so and the whole-network executor (ru_unet_*).
//
// The executor walks the topology of model.UNet (model.py:309-433) once per call and enqueues every kernel on
// the caller's stream; all scratch (activations kept for backward, packed weights, reduction partials) is
// carved out of the caller's workspace by a bump allocator whose layout is a pure function of the shapes, so
// the same walk in "dry" mode is the workspace-size query.  No hipMalloc, no synchronisation.
//
// Forward dataflow of one Residual block (model.py:99-117), C channels:
//   x --conv3--> y1 (+tile stats) --finalize--> (scale1, shift1)
//   y1 --conv3 with fused lrelu(y1*scale1+shift1) on load--> y2 (+tile stats) --finalize--> (scale2, shift2)
//   out = x + lrelu(y2*scale2+shift2)                               (one streaming pass)
// i.e. the GroupNorm-apply + LeakyReLU between the two convolutions never touches HBM.
#include "ru_common.h"

#include <vector>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

namespace ru {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return RU_EHIP;
}

constexpr float kSlope = 1e-2f;   // LeakyReLU(1e-2), model.py:93-94,352
constexpr int kGroups = 8;        // GroupNorm(8, C), model.py:95-96,338
constexpr float kEps = 1e-5f;
constexpr int kInCh = 4;          // model.py:336

// Bump allocator over the caller's workspace.  Tensors grow from the bottom (`off`); an inference forward REWINDS to a mark when a
// block's temporaries are dead (everything runs on one stream, so a later kernel may overwrite them), `peak` is the high-water mark.
// Small per-layer arrays that must outlive those rewinds (GroupNorm mean / rstd / scale / shift, read by ru_unet_gn_stats and by the
// backward) are taken from the TOP of the workspace (`keep`).  The dry walk of ru_unet_workspace_bytes mirrors the real one call for call.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0, keep = 0;
    bool dry = true, failed = false;
    float* alloc(size_t nfloats) {
        const size_t bytes = align_up(nfloats * sizeof(float), 256);
        const size_t o = off;
        off += bytes;
        if (off > peak) peak = off;
        if (dry) return nullptr;
        if (off + keep > (cap & ~(size_t)255)) { failed = true; return nullptr; }
        return reinterpret_cast<float*>(base + o);
    }
    float* alloc_keep(size_t nfloats) {
        keep += align_up(nfloats * sizeof(float), 256);
        if (dry) return nullptr;
        const size_t top = cap & ~(size_t)255;           // (a caller may pass any size: the top region stays 256-byte aligned)
        if (off + keep > top) { failed = true; return nullptr; }
        return reinterpret_cast<float*>(base + top - keep);
    }
    void rewind(size_t mark) { off = mark; }
    size_t need() const { return peak + keep; }
};

// RU_TRACE=1 in the environment: every launch of the executor is named on stderr and followed by a stream synchronisation, so a
// faulting kernel is the last line printed (debugging aid; one getenv per process, nothing on the normal path)
static bool trace_on() {
    static const bool on = [] { const char* e = getenv("RU_TRACE"); return e && *e && *e != '0'; }();
    return on;
}
static int trace_sync(const char* what, hipStream_t s) {
    fprintf(stderr, "[ru] %s\n", what);
    const hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) { fprintf(stderr, "[ru]   -> %s\n", hipGetErrorString(e)); return hip_fail(e, what); }
    return RU_OK;
}

// ru_unet_probe(h, 2): every launch of the executor is bracketed by a HIP event pair on the launch stream and booked to a kernel FAMILY
// (bench.py's `roofline_families`): the family follows from the launcher's name, the level (16-channel level or deeper) from a hint the
// block walkers set.  Thread-local: the sink of the handle whose ru_unet_forward / ru_unet_backward is executing on this thread.
enum { FAM_CONV_L0 = 0, FAM_CONV_DEEP, FAM_WGRAD_L0, FAM_WGRAD_DEEP, FAM_GN, FAM_PW, FAM_OTHER, FAM_COUNT };
// ... and, for the launches that are ONE kernel instantiation worth naming (bench.py's `roofline_top`: the largest single rows of the step's kernel table), to
// an INSTANCE as well: the block walkers tag the launch they are about to make (t_hint_inst, consumed by the next RU_RUN)
enum { INST_CONV16_FWD = 0, INST_CONV16_DGRAD, INST_CONV_DEEP_FWD, INST_CONV_DEEP_DGRAD, INST_WGRAD16_FUSED, INST_WGRAD16_PLAIN, INST_WGRAD_DEEP, INST_COUNT };
struct FamilySink {
    std::vector<hipEvent_t> ev;            // pairs, reused
    std::vector<int> fam;                  // family of pair i
    std::vector<int> inst;                 // instance of pair i (-1: none)
    size_t used = 0;
    ~FamilySink() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
};
static thread_local FamilySink* t_sink = nullptr;
static thread_local int t_hint_c = 16;     // channel count of the level being walked (conv / weight-gradient family split)
static thread_local int t_hint_inst = -1;  // instance of the NEXT launch (INST_*), reset by it
// RU_FUSE_BATCH_WREDUCE: the running backward's queues of weight-gradient reductions -- one per stream the partials are produced on
// (the side stream's are flushed on the side stream before the join, in the shadow of the chain; the caller's stream's at the end)
struct RedQueues { WgradRedList main, side; hipStream_t side_stream = nullptr; };
static thread_local RedQueues* t_red = nullptr;
static WgradRedList* red_for(hipStream_t s) { return t_red ? ((t_red->side_stream && s == t_red->side_stream) ? &t_red->side : &t_red->main) : nullptr; }
static int family_of(const char* call) {
    auto has = [&](const char* k) { return strstr(call, k) != nullptr; };
    if (has("wgrad3_launch")) return t_hint_c <= 16 ? FAM_WGRAD_L0 : FAM_WGRAD_DEEP;
    if (has("wgrad_reduce_flush")) return FAM_WGRAD_DEEP;
    if (has("conv3_launch") || has("conv3_sb_launch")) return t_hint_c <= 16 ? FAM_CONV_L0 : FAM_CONV_DEEP;
    if (has("gn_")) return FAM_GN;
    if (has("conv1_") || has("wgrad1_launch") || has("up2_") || has("s2d_launch") || has("d2s_launch") || has("lrelu_bwd_launch")) return FAM_PW;
    return FAM_OTHER;
}
static void sink_begin(const char* call, hipStream_t s) {
    FamilySink& k = *t_sink;
    while (k.ev.size() < k.used + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        k.ev.push_back(e);
    }
    if (k.fam.size() < k.ev.size() / 2) { k.fam.resize(k.ev.size() / 2); k.inst.resize(k.ev.size() / 2); }
    k.fam[k.used / 2] = family_of(call);
    k.inst[k.used / 2] = t_hint_inst;
    t_hint_inst = -1;
    (void)hipEventRecord(k.ev[k.used], s);
}
static void sink_end(hipStream_t s) {
    FamilySink& k = *t_sink;
    if (k.ev.size() < k.used + 2) return;
    (void)hipEventRecord(k.ev[k.used + 1], s);
    k.used += 2;
}

#define RU_RUN(call)                      \
    do {                                  \
        if (!A.dry) {                     \
            if (A.failed) { set_error("workspace too small"); return RU_ENOMEM; } \
            if (t_sink) sink_begin(#call, s); \
            const int rc__ = (call);      \
            if (t_sink) sink_end(s);      \
            if (rc__ != RU_OK) return rc__; \
            if (trace_on()) { const int rt__ = trace_sync(#call, s); if (rt__ != RU_OK) return rt__; } \
        }                                 \
    } while (0)

__global__ void gn_scale_shift_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                      const float* __restrict__ rstd, float* __restrict__ scale, float* __restrict__ shift, int N, int C, int G) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i % C, g = c / (C / G);
    const float a = gamma[c] * rstd[n * G + g];
    scale[i] = a;
    shift[i] = beta[c] - mean[n * G + g] * a;
}

}  // namespace ru

using namespace ru;

// ====================================================================== engine
struct ParamInfo {
    std::string name;
    int ndim;
    int dims[5];
    size_t offset, numel;
    bool dead;
};

struct BlockP {           // parameter indices of one Residual (model.py:81-97)
    int down = -1, conv1 = -1, conv2 = -1, n1w = -1, n1b = -1, n2w = -1, n2b = -1;
    int cin_down = 0, c = 0;
    // packed-weight slots (offsets in floats into the pack region)
    size_t pk_f1 = 0, pk_f2 = 0, pk_d1 = 0, pk_d2 = 0, pk_downT = 0, pk_down16 = 0;   // pk_down16: [C][8*cin] then [8*cin][C]
    size_t fk_f1 = 0, fk_f2 = 0, fk_d1 = 0, fk_d2 = 0;     // byte offsets of the split-bf16 weight fragments
};

struct GNSave {
    float *mean = nullptr, *rstd = nullptr, *scale = nullptr, *shift = nullptr;
    float act_slope = 0.01f;        // LeakyReLU slope that follows this GroupNorm (norm_input has none: 1)
    float* k = nullptr;             // training, voxel-major flow: constants of the fused GroupNorm-backward statistics (Conv3Args::bst_k)
};

struct BlockSave {
    const BlockP* bp = nullptr;
    const float* xprev = nullptr;   // input before the optional down-sampling conv
    float* xs2d = nullptr;          // space-to-depth of xprev (down blocks)
    const float* x = nullptr;       // block input (after down-sampling)
    const GNSave* xg = nullptr;     // ... which is GroupNorm(+activation) of x applied on the fly when set (first block after the stem)
    float *y1 = nullptr, *y2 = nullptr, *out = nullptr;
    GNSave g1, g2;
    int N = 0, C = 0, D = 0, H = 0, W = 0;   // extents of x / y / out
};

struct DecSave {
    const float* z = nullptr;      // coarse input
    float *u = nullptr, *v = nullptr, *c = nullptr;
    const float* skip = nullptr;
    int level = 0;
};

constexpr int kOneProduct = 0x100;         // flag in wgrad3_run's mode argument
struct ru_unet {
    int depth = 0, nout = 0;
    std::vector<int> enc, dec, ch;
    std::vector<ParamInfo> params;
    size_t total = 0;
    std::vector<std::vector<BlockP>> enc_blocks, dec_blocks;
    std::vector<BlockP> first_blocks;
    std::vector<int> up_w, dec1_w;
    std::vector<size_t> pk_upT, pk_decT;
    int conv_in = -1, nin_w = -1, nin_b = -1, conv_out_w = -1, conv_out_b = -1;
    size_t pk_in = 0, pk_out = 0, pk_out_d = 0, pk_total = 0;
    size_t fk_in = 0, fk_out = 0, fk_out_d = 0, fk_total = 0;
    int precision = RU_PREC_F32;
    unsigned fusion = RU_FUSE_GN_BWD_STATS | RU_FUSE_GN_BWD_APPLY | RU_FUSE_SIDE_STREAM | RU_FUSE_BATCH_WREDUCE | RU_FUSE_PW_DGRAD;     // (RU_FUSE_TAIL_FINALIZE: opt-in)
    int grad_precision = RU_PREC_BF16X3;   // ru_unet_set_grad_precision: RU_PREC_BF16 = one MFMA product in the 3x3x3 data / weight gradients
    int grad_products() const { return (precision == RU_PREC_BF16X3 && grad_precision == RU_PREC_BF16) ? 1 : 3; }
    int wgrad_mode() const { return precision | (grad_products() == 1 ? kOneProduct : 0); }   // `mode` argument of wgrad3_run
    bool c16 = false;           // this forward/backward pair keeps its activations voxel-major (split-bf16, channels % 16 == 0)
    char* fpack = nullptr;

    // state of the last forward
    bool have_fwd = false, training = false;
    int N = 0, D = 0, H = 0, W = 0;
    char* ws = nullptr;
    size_t ws_bytes = 0, fwd_end = 0, fwd_keep = 0;
    float* pack = nullptr;
    const float* x_in = nullptr;
    const float* x_in4 = nullptr;   // 4-channel copy of the input made for the stem conv (C16 flow), reused by its weight gradient
    bool x_in4_planned = false;     // ... decided by structure (the dry walk has null pointers)
    // ru_unet_freeze_params: the weight packs at the head of the workspace are reused while (params, workspace, precision, layout) match
    bool params_frozen = false;
    const float* packed_params = nullptr;
    const void* packed_base = nullptr;
    int packed_prec = -1;
    bool packed_c16 = false;
    int pack_sig = -1;                   // conv3_sb_switch_signature() when the packs of the last forward were written
    float *y0 = nullptr, *t0 = nullptr, *probs = nullptr;
    GNSave g0;
    const float* head_in = nullptr;
    std::vector<BlockSave> first_s;
    std::vector<std::vector<BlockSave>> enc_s, dec_s;
    std::vector<DecSave> dstage;
    std::vector<const float*> skips;
    std::vector<GNSave> gn_order;
    // ru_unet_probe: HIP event pairs around the launches of the dominant kernel (3x3x3 conv 16->16 at the input resolution, forward)
    bool probe_on = false;
    bool probe_families = false;           // ru_unet_probe(h, 2): every launch, booked per kernel family
    ru::FamilySink* sink = nullptr;        // (owned; created on demand)
    std::vector<hipEvent_t> probe_ev;      // pairs (begin, end), created on demand, reused
    size_t probe_used = 0;                 // events recorded since the last read
    // RU_FUSE_SIDE_STREAM: the 3x3x3 weight gradients of the levels below the first (nobody reads them before the optimizer) go to a second,
    // lower-priority HIP stream, event-ordered behind the kernel that produces their dy; ru_unet_backward joins it before it returns
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> fork_ev;       // created on demand (no timing), reused every step
    ru::RedQueues red;                     // RU_FUSE_BATCH_WREDUCE: reductions queued by the running backward
    // RU_FUSE_TAIL_FINALIZE: arrival tickets of the launches that finalize their own partial sums (FinTail).  256 words of device memory
    // owned by the handle, zeroed once when they are created (first forward, outside any capture); the finisher of a launch resets
    // its word, so a step -- eager or replayed from a hipGraph -- always finds zeros.  Every launch of a forward / backward pair takes its own.
    unsigned* tickets = nullptr;
    int ticket_next = 0;
    unsigned* next_ticket() { return tickets ? tickets + (ticket_next++ & 255) : nullptr; }
    bool tails() const { return c16 && (fusion & RU_FUSE_TAIL_FINALIZE) && tickets != nullptr; }
    size_t fork_used = 0;
    ~ru_unet() {
        for (hipEvent_t e : probe_ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : fork_ev) (void)hipEventDestroy(e);
        if (side) (void)hipStreamDestroy(side);
        if (tickets) (void)hipFree(tickets);
        delete sink;
    }
};

static int add_param(ru_unet* h, const std::string& name, std::initializer_list<int> dims, bool dead = false) {
    ParamInfo p;
    p.name = name;
    p.ndim = (int)dims.size();
    size_t n = 1;
    int i = 0;
    for (int d : dims) { p.dims[i++] = d; n *= (size_t)d; }
    for (; i < 5; ++i) p.dims[i] = 1;
    p.offset = h->total;
    p.numel = n;
    p.dead = dead;
    h->total += n;
    h->params.push_back(p);
    return (int)h->params.size() - 1;
}

static BlockP add_block(ru_unet* h, const std::string& prefix, int cin_down, int c, bool dead = false) {
    BlockP b;
    b.c = c;
    b.cin_down = cin_down;
    if (cin_down > 0) b.down = add_param(h, prefix + "downsample.0.weight", {c, cin_down, 2, 2, 2}, dead);
    b.conv1 = add_param(h, prefix + "conv1.conv1.weight", {c, c, 3, 3, 3}, dead);
    b.conv2 = add_param(h, prefix + "conv2.conv1.weight", {c, c, 3, 3, 3}, dead);
    b.n1w = add_param(h, prefix + "norm1.weight", {c}, dead);
    b.n1b = add_param(h, prefix + "norm1.bias", {c}, dead);
    b.n2w = add_param(h, prefix + "norm2.weight", {c}, dead);
    b.n2b = add_param(h, prefix + "norm2.bias", {c}, dead);
    return b;
}

static std::string fmt(const char* f, int a, int b = 0) {
    char buf[128];
    snprintf(buf, sizeof(buf), f, a, b);
    return buf;
}

static void assign_block_packs(ru_unet* h, BlockP& b) {
    const size_t n = conv3_packed_floats(b.c, b.c);
    b.pk_f1 = h->pk_total; h->pk_total += n;
    b.pk_f2 = h->pk_total; h->pk_total += n;
    b.pk_d1 = h->pk_total; h->pk_total += n;
    b.pk_d2 = h->pk_total; h->pk_total += n;
    if (b.down >= 0) {
        b.pk_downT = h->pk_total; h->pk_total += (size_t)8 * b.cin_down * b.c;
        b.pk_down16 = h->pk_total; h->pk_total += (size_t)2 * 8 * b.cin_down * b.c;
    }
    const size_t f = conv3_sb_frag_bytes(b.c, b.c);
    b.fk_f1 = h->fk_total; h->fk_total += f;
    b.fk_f2 = h->fk_total; h->fk_total += f;
    b.fk_d1 = h->fk_total; h->fk_total += f;
    b.fk_d2 = h->fk_total; h->fk_total += f;
}

extern "C" ru_unet_t ru_unet_create(int depth, const int* encoder_layers, const int* decoder_layers,
                                    const int* number_of_channels, int number_of_outputs) {
    if (depth < 2 || depth > 8 || !encoder_layers || !decoder_layers || !number_of_channels || number_of_outputs < 1) {
        set_error("ru_unet_create: bad configuration");
        return nullptr;
    }
    for (int i = 0; i < depth; ++i) {
        if (number_of_channels[i] % kGroups != 0 || number_of_channels[i] <= 0 || encoder_layers[i] < 1 || decoder_layers[i] < 1) {
            set_error("ru_unet_create: channels must be positive multiples of 8 (GroupNorm(8, C)) and layer counts >= 1");
            return nullptr;
        }
    }
    ru_unet* h = new ru_unet();
    if (const char* e = getenv("RU_SIDE_STREAM")) { if (*e == '0') h->fusion &= ~(unsigned)RU_FUSE_SIDE_STREAM; }    // same-box A/B of the side stream
    if (const char* e = getenv("RU_FUSION_OFF")) h->fusion &= ~(unsigned)strtoul(e, nullptr, 0);                   // same-box A/B of any fusion bit (RU_FUSE_*)
    if (const char* e = getenv("RU_FUSION_ON")) h->fusion |= (unsigned)strtoul(e, nullptr, 0) & 63u;
    h->depth = depth;
    h->nout = number_of_outputs;
    h->enc.assign(encoder_layers, encoder_layers + depth);
    h->dec.assign(decoder_layers, decoder_layers + depth);
    h->ch.assign(number_of_channels, number_of_channels + depth);
    const std::vector<int>& ch = h->ch;
    // reference state_dict() order (model.py:320-357): encoder_convs, upsampling, decoder_convs, decoder_convs1x1,
    // conv_input, norm_input, conv_first, conv_output
    h->enc_blocks.resize(depth - 1);
    for (int i = 0; i < depth - 1; ++i)
        for (int j = 0; j < h->enc[i + 1]; ++j)
            h->enc_blocks[i].push_back(add_block(h, fmt("encoder_convs.%d.%d.", i, j), j == 0 ? ch[i] : 0, ch[i + 1]));
    for (int i = 0; i < depth - 1; ++i) h->up_w.push_back(add_param(h, fmt("upsampling.%d.1.weight", i), {ch[i], ch[i + 1], 1, 1, 1}));
    h->dec_blocks.resize(depth);
    for (int i = 0; i < depth; ++i)
        for (int j = 0; j < h->dec[i]; ++j)
            h->dec_blocks[i].push_back(add_block(h, fmt("decoder_convs.%d.%d.", i, j), 0, ch[i], i == depth - 1));
    for (int i = 0; i < depth; ++i)
        h->dec1_w.push_back(add_param(h, fmt("decoder_convs1x1.%d.weight", i), {ch[i], 2 * ch[i], 1, 1, 1}, i == depth - 1));
    h->conv_in = add_param(h, "conv_input.weight", {ch[0], kInCh, 3, 3, 3});
    h->nin_w = add_param(h, "norm_input.weight", {ch[0]});
    h->nin_b = add_param(h, "norm_input.bias", {ch[0]});
    for (int j = 0; j < h->enc[0]; ++j) h->first_blocks.push_back(add_block(h, fmt("conv_first.%d.", j), 0, ch[0]));
    h->conv_out_w = add_param(h, "conv_output.weight", {number_of_outputs, ch[0], 3, 3, 3});
    h->conv_out_b = add_param(h, "conv_output.bias", {number_of_outputs});
    // packed-weight region layout
    h->pk_total = 0;
    h->pk_in = h->pk_total; h->pk_total += conv3_packed_floats(kInCh, ch[0]);
    h->pk_out = h->pk_total; h->pk_total += conv3_packed_floats(ch[0], number_of_outputs);
    h->pk_out_d = h->pk_total; h->pk_total += conv3_packed_floats(number_of_outputs, ch[0]);
    h->fk_total = 0;
    h->fk_in = h->fk_total; h->fk_total += conv3_sb_frag_bytes(kInCh, ch[0]);
    h->fk_out = h->fk_total; h->fk_total += conv3_sb_frag_bytes(ch[0], number_of_outputs);
    h->fk_out_d = h->fk_total; h->fk_total += conv3_sb_frag_bytes(number_of_outputs, ch[0]);
    for (auto& b : h->first_blocks) assign_block_packs(h, b);
    for (auto& lv : h->enc_blocks) for (auto& b : lv) assign_block_packs(h, b);
    for (int i = 0; i < depth - 1; ++i) for (auto& b : h->dec_blocks[i]) assign_block_packs(h, b);
    for (int i = 0; i < depth - 1; ++i) {
        h->pk_upT.push_back(h->pk_total); h->pk_total += (size_t)ch[i + 1] * ch[i];
        h->pk_decT.push_back(h->pk_total); h->pk_total += (size_t)2 * ch[i] * ch[i];
    }
    return h;
}

extern "C" void ru_unet_destroy(ru_unet_t h) { delete h; }
extern "C" int ru_unet_set_precision(ru_unet_t h, int precision) {
    RU_REQUIRE(h && (precision == RU_PREC_F32 || precision == RU_PREC_BF16X3), "ru_unet_set_precision: bad argument");
    h->precision = precision;
    h->have_fwd = false;
    return RU_OK;
}
extern "C" int ru_unet_get_precision(ru_unet_t h) { return h ? h->precision : -1; }
extern "C" int ru_unet_set_fusion(ru_unet_t h, unsigned mask) {
    RU_REQUIRE(h && (mask & ~(unsigned)(RU_FUSE_GN_BWD_STATS | RU_FUSE_GN_BWD_APPLY | RU_FUSE_SIDE_STREAM | RU_FUSE_BATCH_WREDUCE | RU_FUSE_TAIL_FINALIZE | RU_FUSE_PW_DGRAD)) == 0, "ru_unet_set_fusion: bad argument");
    if ((mask & RU_FUSE_TAIL_FINALIZE) && !h->tickets) {      // here, not lazily in ru_unet_forward: that entry must stay free of allocations (stream capture)
        hipError_t e = hipMalloc((void**)&h->tickets, 256 * sizeof(unsigned));
        if (e != hipSuccess) { h->tickets = nullptr; (void)hipGetLastError(); }      // no device in this process (layout-only use): the first forward allocates
    }
    h->fusion = mask;
    h->have_fwd = false;            // the workspace layout of the backward depends on it
    return RU_OK;
}
extern "C" int ru_unet_probe(ru_unet_t h, int enable) {
    RU_REQUIRE(h, "ru_unet_probe: null handle");
    h->probe_on = enable == 1;
    h->probe_families = enable == 2;
    h->probe_used = 0;
    if (h->probe_families && !h->sink) h->sink = new ru::FamilySink();
    if (h->sink) h->sink->used = 0;
    return RU_OK;
}
extern "C" int ru_unet_probe_read_families(ru_unet_t h, double* ms, int* launches, int nfam) {
    RU_REQUIRE(h && ms && launches && (nfam == ru::FAM_COUNT || nfam == ru::FAM_COUNT + ru::INST_COUNT),
               "ru_unet_probe_read_families: %d families (+ %d instance rows behind them)", (int)ru::FAM_COUNT, (int)ru::INST_COUNT);
    for (int f = 0; f < nfam; ++f) { ms[f] = 0.0; launches[f] = 0; }
    if (!h->sink) return RU_OK;
    ru::FamilySink& k = *h->sink;
    for (size_t i = 0; i + 1 < k.used; i += 2) {
        hipError_t e = hipEventSynchronize(k.ev[i + 1]);
        if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize(family probe)");
        float t = 0.f;
        e = hipEventElapsedTime(&t, k.ev[i], k.ev[i + 1]);
        if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime(family probe)");
        ms[k.fam[i / 2]] += t;
        launches[k.fam[i / 2]] += 1;
        const int in = k.inst[i / 2];
        if (in >= 0 && ru::FAM_COUNT + in < nfam) { ms[ru::FAM_COUNT + in] += t; launches[ru::FAM_COUNT + in] += 1; }
    }
    k.used = 0;
    return RU_OK;
}
extern "C" int ru_unet_probe_read(ru_unet_t h, double* total_ms, int* launches) {
    RU_REQUIRE(h && total_ms && launches, "ru_unet_probe_read: null argument");
    double sum = 0.0;
    for (size_t i = 0; i + 1 < h->probe_used; i += 2) {
        hipError_t e = hipEventSynchronize(h->probe_ev[i + 1]);
        if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize(probe)");
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, h->probe_ev[i], h->probe_ev[i + 1]);
        if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime(probe)");
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int)(h->probe_used / 2);
    h->probe_used = 0;
    return RU_OK;
}
extern "C" int ru_unet_set_grad_precision(ru_unet_t h, int precision) {
    RU_REQUIRE(h && (precision == RU_PREC_BF16X3 || precision == RU_PREC_BF16), "ru_unet_set_grad_precision: RU_PREC_BF16X3 or RU_PREC_BF16");
    h->grad_precision = precision;
    return RU_OK;
}
extern "C" int ru_unet_get_grad_precision(ru_unet_t h) { return h ? h->grad_precision : -1; }
extern "C" int ru_unet_freeze_params(ru_unet_t h, int frozen) {
    RU_REQUIRE(h, "ru_unet_freeze_params: null handle");
    h->params_frozen = frozen != 0;
    h->packed_params = nullptr;                          // the next forward packs once, later ones reuse
    return RU_OK;
}
extern "C" int ru_unet_param_count(ru_unet_t h) { return h ? (int)h->params.size() : 0; }
extern "C" const char* ru_unet_param_name(ru_unet_t h, int i) { return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].name.c_str() : nullptr; }
extern "C" int ru_unet_param_ndim(ru_unet_t h, int i) { return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].ndim : 0; }
extern "C" int ru_unet_param_dim(ru_unet_t h, int i, int d) { return (h && i >= 0 && i < (int)h->params.size() && d >= 0 && d < 5) ? h->params[i].dims[d] : 0; }
extern "C" size_t ru_unet_param_offset(ru_unet_t h, int i) { return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].offset : 0; }
extern "C" size_t ru_unet_param_total(ru_unet_t h) { return h ? h->total : 0; }
extern "C" int ru_unet_param_is_dead(ru_unet_t h, int i) { return (h && i >= 0 && i < (int)h->params.size()) ? (int)h->params[i].dead : 0; }

// ---------------------------------------------------------------------- forward pieces
static inline const float* P(const ru_unet* h, const float* params, int idx) { return params ? params + h->params[idx].offset : nullptr; }
static inline float* G(const ru_unet* h, float* grads, int idx) { return grads ? grads + h->params[idx].offset : nullptr; }

static int pack_all(ru_unet* h, const float* params, Arena& A, hipStream_t s) {
    float* pk = h->pack;
    SbPackBatch batch;
    batch.n = 0;
    // one 3x3x3 weight -> the layout of the active precision (f32: K-major floats; bf16x3: hi/lo fragments, all in one launch)
    // lvl >= 0 (training, voxel-major flow): the resolution level whose forward convolution reads this weight -- where that launch takes the Winograd-z / MX kernel
    // of its shape, the direct fragments are not written
    auto pack3 = [&](int pidx, size_t pk_off, size_t fk_off, int cin_f, int cout_f, int mode, int lvl = -1) -> int {
        const bool skipd = h->training && h->c16 && mode == 0 && lvl >= 0 &&
                           conv3_sb_forward_skips_direct(h->N, cin_f, cout_f, h->D >> lvl, h->H >> lvl, h->W >> lvl);
        if (h->precision == RU_PREC_BF16X3) RU_RUN(conv3_sb_pack_add(batch, P(h, params, pidx), h->fpack + fk_off, cin_f, cout_f, mode, !h->training, s, skipd));
        // exact-f32 voxel-major inference: per-lane f32 fragments in the same slot (never larger than the split-bf16 ones)
        if (h->precision == RU_PREC_F32 && h->c16) RU_RUN(conv3_f32c_pack_weights(P(h, params, pidx), h->fpack + fk_off, cin_f, cout_f, mode, s));
        // outside the voxel-major flow the f32 layout is always kept: ragged W falls back to the f32 kernel
        if (!h->c16) RU_RUN(conv3_pack_weights(P(h, params, pidx), pk + pk_off, cin_f, cout_f, mode, s));
        return RU_OK;
    };
    auto blk = [&](const BlockP& b, int lvl) -> int {
        int rc = pack3(b.conv1, b.pk_f1, b.fk_f1, b.c, b.c, 0, lvl); if (rc) return rc;
        rc = pack3(b.conv2, b.pk_f2, b.fk_f2, b.c, b.c, 0, lvl); if (rc) return rc;
        if (h->training) {
            rc = pack3(b.conv1, b.pk_d1, b.fk_d1, b.c, b.c, 1); if (rc) return rc;
            rc = pack3(b.conv2, b.pk_d2, b.fk_d2, b.c, b.c, 1); if (rc) return rc;
        }
        if (b.down >= 0) {
            if (h->c16) RU_RUN(pack_down16_launch(P(h, params, b.down), pk + b.pk_down16, pk + b.pk_down16 + (size_t)8 * b.cin_down * b.c, b.c, b.cin_down, s));
            else RU_RUN(transpose_launch(P(h, params, b.down), pk + b.pk_downT, b.c, 8 * b.cin_down, s));
        }
        return RU_OK;
    };
    { int rc = pack3(h->conv_in, h->pk_in, h->fk_in, kInCh, h->ch[0], 0); if (rc) return rc; }
    { int rc = pack3(h->conv_out_w, h->pk_out, h->fk_out, h->ch[0], h->nout, 0); if (rc) return rc; }
    if (h->training) { int rc = pack3(h->conv_out_w, h->pk_out_d, h->fk_out_d, h->ch[0], h->nout, 1); if (rc) return rc; }
    for (auto& b : h->first_blocks) { int rc = blk(b, 0); if (rc) return rc; }
    for (size_t lv = 0; lv < h->enc_blocks.size(); ++lv) for (auto& b : h->enc_blocks[lv]) { int rc = blk(b, (int)lv + 1); if (rc) return rc; }
    for (int i = 0; i < h->depth - 1; ++i) for (auto& b : h->dec_blocks[i]) { int rc = blk(b, i); if (rc) return rc; }
    for (int i = 0; i < h->depth - 1; ++i) {
        if (h->c16 && !h->training) continue;             // the C16 forward reads the 1x1 weights as stored; the transposes feed its backward only
        RU_RUN(transpose_launch(P(h, params, h->up_w[i]), pk + h->pk_upT[i], h->ch[i], h->ch[i + 1], s));
        RU_RUN(transpose_launch(P(h, params, h->dec1_w[i]), pk + h->pk_decT[i], h->ch[i], 2 * h->ch[i], s));
    }
    RU_RUN(conv3_sb_pack_batch(batch, s));
    return RU_OK;
}

// y = conv3(x) with optional fused input transform, tile statistics -> GNSave (mean/rstd/scale/shift)
static int conv3_gn(ru_unet* h, Arena& A, hipStream_t s, const float* x, const float* wp, const char* wf, float* y, const GNSave* in_gn,
                    const float* gamma, const float* beta, GNSave& out_gn, int N, int Cin, int Cout, int D, int H, int W, bool x_c16 = true,
                    bool x_c4 = false) {
    const int nblk = !h->c16 ? conv3_tiles_per_sample(N, Cin, Cout, D, H, W, h->precision)
                     : (h->precision == RU_PREC_F32 ? conv3_f32c_tiles_per_sample(N, Cin, Cout, D, H, W) : conv3_sb_tiles_per_sample(N, Cin, Cout, D, H, W));
    t_hint_c = Cout;
    float* partials = A.alloc((size_t)N * Cout * nblk * 2);
    out_gn.mean = A.alloc_keep((size_t)N * kGroups);
    out_gn.rstd = A.alloc_keep((size_t)N * kGroups);
    out_gn.scale = A.alloc_keep((size_t)N * Cout);
    out_gn.shift = A.alloc_keep((size_t)N * Cout);
    Conv3Args a{};
    a.x = x; a.wp = wp; a.y = y; a.mode = h->precision; a.wfrag = wf;
    a.in_scale = in_gn ? in_gn->scale : nullptr;
    a.in_shift = in_gn ? in_gn->shift : nullptr;
    a.in_slope = in_gn ? in_gn->act_slope : kSlope;
    a.stat_partials = partials;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    a.in_c16 = h->c16 && x_c16; a.out_c16 = h->c16; a.in_c4 = x_c4;
    a.products = 2;          // a FORWARD convolution: its input is an activation tensor, so the shapes that have the kernel take the fp16 + MX-fp8 scheme (conv3_mx.hpp; RU_MX=0: never)
    const bool probed = h->probe_on && !A.dry && h->c16 && x_c16 && !x_c4 && Cin == 16 && Cout == 16 && D == h->D && H == h->H && W == h->W;
    if (probed) {
        while (h->probe_ev.size() < h->probe_used + 2) {
            hipEvent_t e;
            const hipError_t er = hipEventCreate(&e);
            if (er != hipSuccess) return hip_fail(er, "hipEventCreate(probe)");
            h->probe_ev.push_back(e);
        }
        (void)hipEventRecord(h->probe_ev[h->probe_used], s);
    }
    // exact-f32 engine, small shapes (the deep levels of a batch-1 forward): input-channel chunks split over co-resident workgroups, the
    // partial output tensors summed in a fixed order and the GroupNorm statistics taken by the stand-alone pass
    const int ksplit = (!h->c16 && conv3_effective_mode(h->precision, W) == RU_PREC_F32) ? conv3_f32_ksplit(N, Cin, Cout, D, H, W) : 1;
    out_gn.k = (h->training && h->c16) ? A.alloc_keep((size_t)N * 3 * Cout) : nullptr;
    // RU_FUSE_TAIL_FINALIZE: the conv's last workgroup turns the partials into mean / rstd / scale / shift (every split-bf16 kernel has the tail)
    const bool tail = !A.dry && h->tails() && ksplit == 1 && h->precision == RU_PREC_BF16X3;
    if (tail) {
        FinTail& f = a.fin;
        f.ticket = h->next_ticket(); f.kind = 1; f.nblk = nblk; f.N = N; f.C = Cout; f.G = kGroups; f.V = (size_t)D * H * W; f.eps = kEps;
        f.gamma = gamma; f.beta = beta; f.mean = out_gn.mean; f.rstd = out_gn.rstd; f.scale = out_gn.scale; f.shift = out_gn.shift; f.bst_k = out_gn.k;
    }
    if (ksplit > 1) {
        const size_t nel = (size_t)N * Cout * D * H * W;
        float* part = A.alloc((size_t)ksplit * nel);
        a.y = part; a.stat_partials = nullptr; a.ksplit = ksplit;
        RU_RUN(conv3_launch(a, s));
        RU_RUN(sum_partials_launch(part, ksplit, nel, y, s));
        RU_RUN(gn_stats_launch(y, partials, N, Cout, (size_t)D * H * W, s));
    } else {
        t_hint_inst = (a.in_c16 && a.out_c16) ? (Cin == 16 && Cout == 16 ? INST_CONV16_FWD : (Cin >= 32 ? INST_CONV_DEEP_FWD : -1)) : -1;
        RU_RUN(conv3_launch(a, s));
    }
    if (probed) {
        (void)hipEventRecord(h->probe_ev[h->probe_used + 1], s);
        h->probe_used += 2;
    }
    if (!tail) RU_RUN(gn_finalize_launch(partials, nblk, gamma, beta, out_gn.mean, out_gn.rstd, out_gn.scale, out_gn.shift, N, Cout,
                                         (size_t)D * H * W, kGroups, kEps, s, out_gn.k));
    h->gn_order.push_back(out_gn);
    return RU_OK;
}

// xin_gn: the block input is GroupNorm(+activation) of the RAW tensor xprev, applied on the fly (voxel-major flow, no down-sampling
// conv): the first conv and its weight gradient stage it with the fused transform, the residual add applies it per element
// defer_out (the block in front of the head conv): the residual pass does not run -- the caller hands (y2, norm2's scale / shift, the block input x) to the
// head conv, whose staging forms x + lrelu(norm2(y2)) itself (Conv3Args::in_res) and, in training, writes it to sv.out on the way (in_sum_out: the head's
// weight gradient reads it); in inference *out stays null and nothing of the block is rewound.
static int block_fwd(ru_unet* h, const float* params, Arena& A, hipStream_t s, const BlockP& bp, const float* xprev,
                     int N, int D, int H, int W /* extents of xprev */, BlockSave& sv, const float** out, const GNSave* xin_gn = nullptr, bool defer_out = false) {
    sv = BlockSave();
    sv.xg = xin_gn;
    sv.bp = &bp;
    sv.xprev = xprev;
    const int C = bp.c;
    const float* x = xprev;
    // inference: nothing but the block's output survives it -- the output is allocated first and the arena rewound behind it at the end
    // (the workspace of 8 tiles of 192^3 fell from 45 to ~20 GiB); training keeps every tensor for the backward
    const bool recycle = !h->training;
    float* out_early = nullptr;
    if (recycle) {
        const int sh = bp.down >= 0 ? 1 : 0;
        out_early = A.alloc((size_t)N * C * (size_t)(D >> sh) * (H >> sh) * (W >> sh));
    }
    const size_t mark = A.off;
    if (bp.down >= 0) {
        const int Do = D / 2, Ho = H / 2, Wo = W / 2;
        const size_t Vo = (size_t)Do * Ho * Wo;
        if (!h->c16) sv.xs2d = A.alloc((size_t)N * 8 * bp.cin_down * Vo);
        float* xd = A.alloc((size_t)N * C * Vo);
        Conv1Args c1{};
        c1.x0 = sv.xs2d; c1.C0 = 8 * bp.cin_down; c1.y = xd; c1.out_slope = 1.f;
        c1.N = N; c1.Cout = C; c1.V = Vo;
        if (h->c16) {                                            // the 8 taps are gathered from xprev: no space-to-depth tensor
            c1.x0 = xprev; c1.s2d = 1; c1.Dc = Do; c1.Hc = Ho; c1.Wc = Wo;
            c1.wT = h->pack + bp.pk_down16; c1.ldw = 8 * bp.cin_down;         // [C][tap*cin + c]
            RU_RUN(conv1_16_launch(c1, s));
        } else {
            RU_RUN(s2d_launch(xprev, sv.xs2d, N, bp.cin_down, D, H, W, s));
            c1.wT = h->pack + bp.pk_downT; c1.ldw = C;
            RU_RUN(conv1_launch(c1, s));
        }
        x = xd; D = Do; H = Ho; W = Wo;
    }
    const size_t V = (size_t)D * H * W;
    sv.x = x; sv.N = N; sv.C = C; sv.D = D; sv.H = H; sv.W = W;
    sv.y1 = A.alloc((size_t)N * C * V);
    int rc = conv3_gn(h, A, s, x, h->pack + bp.pk_f1, h->fpack + bp.fk_f1, sv.y1, xin_gn, P(h, params, bp.n1w), P(h, params, bp.n1b), sv.g1, N, C, C, D, H, W);
    if (rc) return rc;
    sv.y2 = A.alloc((size_t)N * C * V);
    rc = conv3_gn(h, A, s, sv.y1, h->pack + bp.pk_f2, h->fpack + bp.fk_f2, sv.y2, &sv.g1, P(h, params, bp.n2w), P(h, params, bp.n2b), sv.g2, N, C, C, D, H, W);
    if (rc) return rc;
    if (defer_out) {
        RU_REQUIRE(h->c16 && !xin_gn, "block_fwd: the deferred residual pass exists in the voxel-major flow only");
        sv.out = h->training ? A.alloc((size_t)N * C * V) : nullptr;          // training: the head conv's staging WRITES it (Conv3Args::in_sum_out) for the backward
        *out = sv.out;
        return RU_OK;
    }
    sv.out = recycle ? out_early : A.alloc((size_t)N * C * V);
    if (h->c16) RU_RUN(gn_apply16_launch(sv.y2, sv.g2.scale, sv.g2.shift, x, sv.out, N, C, V, kSlope, s,
                                         xin_gn ? xin_gn->scale : nullptr, xin_gn ? xin_gn->shift : nullptr, xin_gn ? xin_gn->act_slope : 1.f));
    else RU_RUN(gn_apply_launch(sv.y2, sv.g2.scale, sv.g2.shift, x, sv.out, N, C, V, kSlope, s));
    *out = sv.out;
    if (recycle) A.rewind(mark);
    return RU_OK;
}

static int unet_forward_impl(ru_unet* h, const float* params, const float* x, float* probs_out, Arena& A, hipStream_t s) {
    const int N = h->N, depth = h->depth;
    std::vector<int> Dl(depth), Hl(depth), Wl(depth);
    for (int i = 0; i < depth; ++i) { Dl[i] = h->D >> i; Hl[i] = h->H >> i; Wl[i] = h->W >> i; }
    auto Vl = [&](int i) { return (size_t)Dl[i] * Hl[i] * Wl[i]; };
    h->gn_order.clear();
    // voxel-major flow: the split-bf16 engine, and (round 5) the exact-f32 INFERENCE forward -- conv3_f32c_kernel on voxel-major tensors with the same
    // fused statistics / staging-side GroupNorm + LeakyReLU / coarse-grid 1x1 / no concat as the split-bf16 flow (BASELINE configs[1]; the exact-f32
    // training path keeps the NCDHW kernels: its weight gradients exist there only).  RU_F32C=0: the NCDHW flow for the f32 forward too (A/B).
    static const bool f32c_off = [] { const char* e = getenv("RU_F32C"); return e && *e == '0'; }();
    // (conv3_f32c_kernel addresses a 16-channel block of its input through 32-bit byte offsets and has no other kernel to fall back to: whole-volume
    // exact-f32 inference at 2^25 voxels and more, ~322^3, keeps the NCDHW flow; the split-bf16 kernels choose the one-stage kernel there themselves)
    const bool f32c_fits = (size_t)h->D * h->H * h->W * 64 < ((size_t)1 << 31);
    h->c16 = (h->precision == RU_PREC_BF16X3 || (h->precision == RU_PREC_F32 && !h->training && !f32c_off && f32c_fits)) && (h->W & 3) == 0;
    for (int c : h->ch) h->c16 = h->c16 && (c % 16 == 0);
    h->pack = A.alloc(h->pk_total);
    h->fpack = reinterpret_cast<char*>(A.alloc(h->fk_total / sizeof(float) + 64));
    float* wf4_in = A.alloc(conv3_sb4_frag_bytes(h->ch[0]) / sizeof(float) + 64);     // stem fragments: next to the packs, so they can be reused with them
    const bool reuse_packs = h->params_frozen && !h->training && !A.dry && h->packed_params == params && h->packed_base == (const void*)h->pack &&
                             h->packed_prec == h->precision && h->packed_c16 == h->c16;
    int rc = RU_OK;
    if (!reuse_packs) {
        rc = pack_all(h, params, A, s);
        if (rc) return rc;
    }

    // stem: conv_input -> norm_input (no activation, model.py:412-413)
    const int C0 = h->ch[0];
    const BlockSave* head_block = nullptr;               // inference: the block whose residual pass the head conv's staging takes over (block_fwd, defer_out)
    h->x_in = x;
    h->x_in4 = nullptr;
    h->x_in4_planned = false;
    h->y0 = A.alloc((size_t)N * C0 * Vl(0));
    const size_t stem_mark = A.off;
    if (h->c16 && h->precision == RU_PREC_BF16X3 && conv3_sb4_usable(N, kInCh, C0, Dl[0], Hl[0], Wl[0])) {
        // few input channels: 4-channel copy + the tap-pair kernel (K = 2 taps x 4 channels per packet) instead of padding 4 -> 16 channels
        float* x4 = A.alloc((size_t)N * 4 * Vl(0));
        float* wf4 = wf4_in;
        h->x_in4 = x4;
        h->x_in4_planned = true;
        RU_RUN(pad_to_c4_launch(x, x4, N, kInCh, Vl(0), s));
        if (!reuse_packs) RU_RUN(conv3_sb4_pack_weights(P(h, params, h->conv_in), wf4, kInCh, C0, 0, s));
        rc = conv3_gn(h, A, s, x4, nullptr, reinterpret_cast<const char*>(wf4), h->y0, nullptr, P(h, params, h->nin_w), P(h, params, h->nin_b), h->g0,
                      N, kInCh, C0, Dl[0], Hl[0], Wl[0], false, true);
    } else {
        rc = conv3_gn(h, A, s, x, h->pack + h->pk_in, h->fpack + h->fk_in, h->y0, nullptr, P(h, params, h->nin_w), P(h, params, h->nin_b), h->g0, N, kInCh, C0, Dl[0], Hl[0], Wl[0], false);
    }
    if (rc) return rc;
    if (!h->training) A.rewind(stem_mark);                      // inference: the 4-channel copy and the statistics partials are dead (the weight gradient reuses the copy in training)
    // norm_input has no activation.  Voxel-major flow: its output is never written -- the first block reads the raw stem output with
    // the affine fused into its staging (conv, weight gradient) and into its residual add
    h->g0.act_slope = 1.0f;
    const bool stem_fused = h->c16 && !h->first_blocks.empty() && h->first_blocks[0].down < 0;
    h->t0 = nullptr;
    if (!stem_fused) {
        h->t0 = A.alloc((size_t)N * C0 * Vl(0));
        if (h->c16) RU_RUN(gn_apply16_launch(h->y0, h->g0.scale, h->g0.shift, nullptr, h->t0, N, C0, Vl(0), 1.0f, s));
        else RU_RUN(gn_apply_launch(h->y0, h->g0.scale, h->g0.shift, nullptr, h->t0, N, C0, Vl(0), 1.0f, s));
    }
    const float* cur = stem_fused ? h->y0 : h->t0;
    h->first_s.assign(h->first_blocks.size(), BlockSave());
    for (size_t j = 0; j < h->first_blocks.size(); ++j) {
        rc = block_fwd(h, params, A, s, h->first_blocks[j], cur, N, Dl[0], Hl[0], Wl[0], h->first_s[j], &cur, (stem_fused && j == 0) ? &h->g0 : nullptr);
        if (rc) return rc;
    }
    // encoder (model.py:416-418)
    h->skips.assign(depth - 1, nullptr);
    h->enc_s.assign(depth - 1, {});
    for (int i = 0; i < depth - 1; ++i) {
        h->skips[i] = cur;
        h->enc_s[i].assign(h->enc_blocks[i].size(), BlockSave());
        for (size_t j = 0; j < h->enc_blocks[i].size(); ++j) {
            const int lv = j == 0 ? i : i + 1;
            rc = block_fwd(h, params, A, s, h->enc_blocks[i][j], cur, N, Dl[lv], Hl[lv], Wl[lv], h->enc_s[i][j], &cur);
            if (rc) return rc;
        }
    }
    // decoder (model.py:420-426)
    h->dstage.assign(depth - 1, DecSave());
    h->dec_s.assign(depth - 1, {});
    for (int i = depth - 2; i >= 0; --i) {
        DecSave& ds = h->dstage[i];
        ds.level = i;
        ds.z = cur;
        ds.skip = h->skips[i];
        const int Ci = h->ch[i], Cc = h->ch[i + 1];
        // C16 flow: the 1x1x1 conv and the trilinear interpolation are both linear and act on different axes, so
        // conv(up(z)) = up(conv(z)): the conv runs on the COARSE grid (8x fewer voxels) and the up-sampling on Ci = Cc/2 channels,
        // with the LeakyReLU fused into its store; the Cc-channel fine tensor `u` never exists.
        const bool recycle = !h->training;                       // inference: only the stage's output c survives it
        float* c_early = recycle ? A.alloc((size_t)N * Ci * Vl(i)) : nullptr;
        const size_t dmark = A.off;
        if (!h->c16) {
            ds.u = A.alloc((size_t)N * Cc * Vl(i));
            RU_RUN(up2_fwd_launch(cur, ds.u, N, Cc, Dl[i + 1], Hl[i + 1], Wl[i + 1], s));
        }
        ds.v = A.alloc((size_t)N * Ci * Vl(i));
        Conv1Args c1{};
        c1.x0 = ds.u; c1.C0 = Cc; c1.y = ds.v; c1.out_slope = kSlope;                                                 // + LeakyReLU (model.py:422)
        c1.N = N; c1.Cout = Ci; c1.V = Vl(i);
        ds.c = recycle ? c_early : A.alloc((size_t)N * Ci * Vl(i));
        Conv1Args c2{};
        c2.x0 = ds.skip; c2.C0 = Ci; c2.x1 = ds.v; c2.C1 = Ci;                                                        // cat([skip, up]) (model.py:424)
        c2.y = ds.c; c2.out_slope = 1.f; c2.N = N; c2.Cout = Ci; c2.V = Vl(i);
        if (h->c16) {                                            // C16 kernel reads the reference layout [out][in] directly
            float* zc = A.alloc((size_t)N * Ci * Vl(i + 1));
            c1.x0 = cur; c1.y = zc; c1.out_slope = 1.f; c1.V = Vl(i + 1);
            c1.wT = P(h, params, h->up_w[i]); c1.ldw = Cc;
            RU_RUN(conv1_16_launch(c1, s));
            RU_RUN(up2_fwd16_launch(zc, ds.v, N, Ci, Dl[i + 1], Hl[i + 1], Wl[i + 1], kSlope, s));
            c2.wT = P(h, params, h->dec1_w[i]); c2.ldw = 2 * Ci;
            RU_RUN(conv1_16_launch(c2, s));
        } else {
            c1.wT = h->pack + h->pk_upT[i]; c1.ldw = Ci;
            RU_RUN(conv1_launch(c1, s));
            c2.wT = h->pack + h->pk_decT[i]; c2.ldw = Ci;
            RU_RUN(conv1_launch(c2, s));
        }
        if (recycle) A.rewind(dmark);
        cur = ds.c;
        h->dec_s[i].assign(h->dec_blocks[i].size(), BlockSave());
        for (size_t j = 0; j < h->dec_blocks[i].size(); ++j) {
            // the block in front of the head conv leaves its residual pass to that conv's staging (inference: one read of y2 and x instead of read y2 +
            // read x + write out + read out, 0.54 GB less per 128^3 volume; training: the staging also writes out, 0.27 GB less)
            const bool defer = i == 0 && j + 1 == h->dec_blocks[i].size() && h->c16 &&
                               (h->precision == RU_PREC_BF16X3 ? conv3_sb_head_takes_residual(N, C0, h->nout, Dl[0], Hl[0], Wl[0])
                                                               : (!h->training && conv3_f32c_head_takes_residual(C0, h->nout, Wl[0])));       // (exact f32: the voxel-major flow is inference only)
            rc = block_fwd(h, params, A, s, h->dec_blocks[i][j], cur, N, Dl[i], Hl[i], Wl[i], h->dec_s[i][j], &cur, nullptr, defer);
            if (rc) return rc;
            if (defer) head_block = &h->dec_s[i][j];
        }
    }
    // head: conv_output + bias + sigmoid (model.py:429-431)
    h->head_in = cur;
    t_hint_c = C0;
    // training: the sigmoid backward reads the probabilities from the CALLER's buffer (kept valid until ru_unet_backward, see the header)
    float* pdst = probs_out;
    h->probs = probs_out;
    Conv3Args a{};
    a.x = cur; a.wp = h->pack + h->pk_out; a.bias = P(h, params, h->conv_out_b); a.y = pdst; a.sigmoid = 1;
    a.mode = h->precision; a.wfrag = h->fpack + h->fk_out; a.in_c16 = h->c16;
    a.N = N; a.Cin = C0; a.Cout = h->nout; a.D = Dl[0]; a.H = Hl[0]; a.W = Wl[0];
    if (head_block) {                                    // x + lrelu(norm2(conv2)) of the last block, formed in this conv's staging
        a.x = head_block->y2; a.in_scale = head_block->g2.scale; a.in_shift = head_block->g2.shift; a.in_slope = kSlope; a.in_res = head_block->x;
        a.in_sum_out = head_block->out;                  // (training: the block output, written on the way; null in inference)
    }
    RU_RUN(conv3_launch(a, s));
    if (!A.dry) { h->packed_params = h->training ? nullptr : params; h->packed_base = h->pack; h->packed_prec = h->precision; h->packed_c16 = h->c16; h->pack_sig = conv3_sb_switch_signature(); }   // a training forward is followed by an optimizer step
    return RU_OK;
}

// ---------------------------------------------------------------------- backward pieces
// GroupNorm(+LeakyReLU) backward: d_act -> dy (gradient w.r.t. the raw conv output), dgamma/dbeta written
// fused_part / fused_nblk: the partial sums were already taken by the conv that produced `dact` (Conv3Args::bst_*): no reduce pass
// pre: the caller allocated the coefficient buffer next to the fused partials (FusedSums) -- and when pre->done, the kernel that produced
// the partials has already finalized them in its tail (RU_FUSE_TAIL_FINALIZE): no finalize launch here
struct FusedSums { float* part = nullptr; int nblk = 0; float* coef = nullptr; bool done = false; };
static void fill_bwd_tail(ru_unet* h, FinTail& f, const FusedSums& fs, const GNSave& g, const float* gamma, float* dgamma, float* dbeta, int N, int C, size_t V, int s2_sign) {
    f.ticket = h->next_ticket(); f.kind = 2; f.nblk = fs.nblk; f.N = N; f.C = C; f.G = kGroups; f.V = V; f.eps = kEps; f.s2_sign = s2_sign;
    f.gamma = gamma; f.mean = g.mean; f.rstd = g.rstd; f.coef = fs.coef; f.dgamma = dgamma; f.dbeta = dbeta;
}
static int gn_bwd(ru_unet* h, Arena& A, hipStream_t s, const float* yraw, const float* dact, const GNSave& g, const float* gamma, float slope,
                  float* dy, float* dgamma, float* dbeta, int N, int C, size_t V, const FusedSums* pre = nullptr,
                  float** coef_out = nullptr /* non-null: stop after the finalize; the apply is fused into the weight gradient (Wgrad3Args::gb_*) */,
                  bool split = true /* voxel-major flow: publish dy as hi/lo bf16 packets (read by the transpose-read weight gradient and the
                                       persistent data-gradient conv) or as plain float32 C16 (the generic weight-gradient kernel) */) {
    const bool c16 = h->c16;
    const bool fused = pre && pre->nblk > 0;
    const int nblk = fused ? pre->nblk : (c16 ? gn_bwd_tiles16(V) : gn_bwd_tiles(V));
    float* part = fused ? pre->part : A.alloc((size_t)N * C * nblk * 2);
    float* coef = fused ? pre->coef : A.alloc((size_t)N * C * 3);
    bool done = fused && pre->done;
    if (!fused) {
        if (c16) {
            FinTail f{};
            if (!A.dry && h->tails()) {
                FusedSums fs; fs.part = part; fs.nblk = nblk; fs.coef = coef;
                fill_bwd_tail(h, f, fs, g, gamma, dgamma, dbeta, N, C, V, 0);
                done = true;
            }
            RU_RUN(gn_bwd_reduce16_launch(yraw, dact, g.scale, g.shift, g.mean, g.rstd, slope, part, N, C, V, kGroups, s, &f));
        } else {
            RU_RUN(gn_bwd_reduce_launch(yraw, dact, g.scale, g.shift, g.mean, g.rstd, slope, part, N, C, V, kGroups, s));
        }
    }
    if (!done) RU_RUN(gn_bwd_finalize_launch(part, nblk, gamma, g.mean, g.rstd, coef, dgamma, dbeta, N, C, V, kGroups, s, fused ? 1 : 0));
    if (coef_out) { *coef_out = coef; return RU_OK; }
    if (c16) RU_RUN(gn_bwd_apply16_launch(yraw, dact, g.scale, g.shift, coef, slope, dy, N, C, V, split ? 1 : 0, s));   // split form: read by MFMA kernels only
    else RU_RUN(gn_bwd_apply_launch(yraw, dact, g.scale, g.shift, coef, slope, dy, N, C, V, s));
    return RU_OK;
}

// GroupNorm-backward apply fused into the weight gradient's dy staging (Wgrad3Args::gb_*): dy is OUTPUT (split form) then
struct GbApply { const float* y; const float* d; const GNSave* g; const float* coef; bool g16 = false; };     // g16: publish the gradient-operand form of the MX scheme (conv3_mxg_usable)
static int wgrad3_run(Arena& A, hipStream_t s, int mode, const float* x, const GNSave* xg, const float* dy, float* dw, int N, int Cin, int Cout, int D, int H, int W,
                      bool x_c16 = false, bool dy_c16 = false, const float* few4 = nullptr, bool dy_s16 = false, const GbApply* gb = nullptr) {
    const int products = (mode & kOneProduct) ? 1 : 3;
    mode &= ~kOneProduct;
    if (x_c16 != dy_c16 && mode == RU_PREC_BF16X3 && Cin <= 16 && Cout <= 16) {
        // stem (x = network input) / head (dy = class gradient): the few-channel NCDHW side enters the transpose-read kernel as a
        // 16-channel block that is zero beyond its real channels -- from the 4-channel copy the conv of that tensor already made
        // (`few4`), else from a zero-padded voxel-major copy made here
        const size_t V = (size_t)D * H * W;
        const int cfew = x_c16 ? Cout : Cin;
        const bool use4 = few4 != nullptr && cfew <= 4;
        float* pad = use4 ? nullptr : A.alloc((size_t)N * 16 * V);
        if (!use4) RU_RUN(pad_to_c16_launch(x_c16 ? dy : x, pad, N, cfew, V, s));
        const float* fewp = use4 ? few4 : pad;
        Wgrad3Args w{};
        w.x = x_c16 ? x : fewp; w.dy = x_c16 ? fewp : dy; w.dw = dw; w.mode = mode; w.x_c16 = 1; w.dy_c16 = 1;
        w.x_c4 = (!x_c16 && use4) ? 1 : 0; w.dy_c4 = (x_c16 && use4) ? 1 : 0; w.dy_s16 = (dy_s16 && !x_c16) ? 1 : 0;
        w.products = products; w.defer = red_for(s);
        // head (dy has 3 real channels, no fused transform on x): exchange the operands, so that the few-channel tensor is the 4-channel
        // x operand whose packet carries the three dx taps (a third of the matrix work); the result comes out transposed with mirrored taps
        const bool swap = x_c16 && use4 && !xg && !gb;
        if (swap) { w.x = fewp; w.dy = x; w.x_c4 = 1; w.dy_c4 = 0; w.dy_s16 = 0; w.swapped = 1; }
        if (gb && !x_c16 && use4) {                              // stem: dy is the GroupNorm-backward apply of norm_input, computed while staging; nobody else reads it
            w.gb_y = gb->y; w.gb_d = gb->d; w.gb_scale = gb->g->scale; w.gb_shift = gb->g->shift; w.gb_coef = gb->coef; w.gb_slope = gb->g->act_slope;
            w.gb_out = nullptr; w.dy = gb->y; w.dy_s16 = 0;      // (dy unused in this mode; any valid pointer)
        }
        w.in_scale = xg ? xg->scale : nullptr; w.in_shift = xg ? xg->shift : nullptr; w.in_slope = xg ? xg->act_slope : kSlope;
        w.dw_cin = Cin; w.dw_cout = Cout;
        w.N = N; w.Cin = x_c16 ? Cin : 16; w.Cout = x_c16 ? 16 : Cout; w.D = D; w.H = H; w.W = W;
        if (swap) { w.dw_cin = Cout; w.dw_cout = Cin; w.Cin = 16; w.Cout = Cin; }      // kernel view: x' = d (few channels), dy' = x
        w.ws_bytes = wgrad3_workspace_bytes(N, w.Cin, w.Cout, D, H, W);
        w.ws = A.alloc(w.ws_bytes / sizeof(float));
        RU_RUN(wgrad3_launch(w, s));
        return RU_OK;
    }
    Wgrad3Args w{};
    w.x = x; w.dy = dy; w.dw = dw; w.mode = mode; w.x_c16 = x_c16; w.dy_c16 = dy_c16; w.dy_s16 = dy_s16 ? 1 : 0;
    w.products = products; w.defer = red_for(s);
    w.in_scale = xg ? xg->scale : nullptr; w.in_shift = xg ? xg->shift : nullptr; w.in_slope = xg ? xg->act_slope : kSlope;
    w.ws_bytes = wgrad3_workspace_bytes(N, Cin, Cout, D, H, W);
    w.ws = A.alloc(w.ws_bytes / sizeof(float));
    w.N = N; w.Cin = Cin; w.Cout = Cout; w.D = D; w.H = H; w.W = W;
    if (gb) {
        w.gb_y = gb->y; w.gb_d = gb->d; w.gb_scale = gb->g->scale; w.gb_shift = gb->g->shift; w.gb_coef = gb->coef; w.gb_slope = kSlope;
        w.gb_out = const_cast<float*>(dy); w.dy_s16 = 0; w.gb_g16 = gb->g16 ? 1 : 0;
    }
    t_hint_inst = (x_c16 && dy_c16) ? (Cin == 16 && Cout == 16 ? (gb ? INST_WGRAD16_FUSED : INST_WGRAD16_PLAIN) : (Cin >= 32 ? INST_WGRAD_DEEP : -1)) : -1;
    RU_RUN(wgrad3_launch(w, s));
    return RU_OK;
}

static int wgrad1_run(Arena& A, hipStream_t s, const float* x, const float* dy, float* dw, int ldw, int N, int Cin, int Cout, size_t V,
                      bool c16 = false, int tap_split = 0) {
    Wgrad1Args w{};
    w.x = x; w.dy = dy; w.dw = dw; w.ldw = ldw; w.c16 = c16; w.tap_split = tap_split; w.defer = red_for(s);
    w.ws_bytes = wgrad1_workspace_bytes(N, Cin, Cout, V);
    w.ws = A.alloc(w.ws_bytes / sizeof(float));
    w.N = N; w.Cin = Cin; w.Cout = Cout; w.V = V;
    RU_RUN(wgrad1_launch(w, s));
    return RU_OK;
}

// Side stream (RU_FUSE_SIDE_STREAM).  side_fork: everything enqueued on `main` so far happens-before what is enqueued on the side stream
// next; side_join: everything enqueued on the side stream so far happens-before what is enqueued on `main` next.  Events only, no host
// synchronisation; the pattern (fork ... join back into the origin stream) is capturable in a hipGraph.
static int side_event(ru_unet* h, hipEvent_t* out) {
    if (h->fork_used == h->fork_ev.size()) {
        hipEvent_t e;
        const hipError_t er = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (er != hipSuccess) return hip_fail(er, "hipEventCreateWithFlags(side)");
        h->fork_ev.push_back(e);
    }
    *out = h->fork_ev[h->fork_used++];
    return RU_OK;
}
static int side_fork(ru_unet* h, hipStream_t main) {
    if (!h->side) {
        int lo = 0, hi = 0;                                      // lowest priority: the chain on `main` is the critical path
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const hipError_t er = hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, lo);
        if (er != hipSuccess) return hip_fail(er, "hipStreamCreateWithPriority(side)");
    }
    h->red.side_stream = h->side;
    hipEvent_t e;
    int rc = side_event(h, &e);
    if (rc) return rc;
    hipError_t er = hipEventRecord(e, main);
    if (er == hipSuccess) er = hipStreamWaitEvent(h->side, e, 0);
    return er == hipSuccess ? RU_OK : hip_fail(er, "side_fork");
}
static int side_join(ru_unet* h, hipStream_t main) {
    if (!h->side || h->fork_used == 0) return RU_OK;
    hipEvent_t e;
    int rc = side_event(h, &e);
    if (rc) return rc;
    hipError_t er = hipEventRecord(e, h->side);
    if (er == hipSuccess) er = hipStreamWaitEvent(main, e, 0);
    h->fork_used = 0;                                            // the events are free again after this step's join
    return er == hipSuccess ? RU_OK : hip_fail(er, "side_join");
}

// Residual backward: dout -> d(xprev); parameter gradients into `grads`
// `join`: a second gradient arriving at the block input (the skip connection of a down block); *joined tells whether it was added here
// part2 / nblk2: the GroupNorm-backward sums of norm2 were taken by the kernel that produced `dout` (Conv3Args::bst_*)
// nx: the GroupNorm(+activation) the gradient this block produces (dx, for a block without down-sampling conv) enters NEXT -- norm2 of the
//     previous block of the level, or norm_input: its backward sums are then taken in the epilogue of the data-gradient conv of conv1
//     (which also adds the skip gradient) and handed back through nx->part / nx->nblk
struct GNNext {
    const float* y; const float* k; float slope;
    const GNSave* g; const float* gamma; float* dgamma; float* dbeta;     // the GroupNorm the produced gradient enters (tail finalize)
    FusedSums out;                                                        // filled here: partials (+ coefficients) of that GroupNorm
};
static int block_bwd(ru_unet* h, const float* params, float* grads, Arena& A, hipStream_t s, const BlockSave& sv, const float* dout,
                     const float** dxprev_out, const float* join = nullptr, bool* joined = nullptr, const FusedSums* sums2 = nullptr,
                     GNNext* nx = nullptr) {
    if (joined) *joined = false;
    const BlockP& bp = *sv.bp;
    const int N = sv.N, C = sv.C, D = sv.D, H = sv.H, W = sv.W;
    t_hint_c = C;
    const size_t V = (size_t)D * H * W;
    float* dy2 = A.alloc((size_t)N * C * V);
    const bool c16 = h->c16;
    // 16-channel level (its kernels are memory-bound): the GroupNorm-backward apply is computed by the weight gradient's dy staging
    // from (y, d, coefficients) and published in split form for the data-gradient conv that follows -- no apply pass over (y, d)
    // (16 channels only.  The two-block kernel wgrad3_tz<2,0,3> exists and fits its registers since the constants of a block are fetched when
    // the block is converted, but at 32 channels it costs what it saves: 203 us against 131 us + a 70 us apply pass, same box,
    // profiles/r02_notes.txt -- every input-channel group recomputes the apply while staging, and the producers become the slower side.)
    const bool fa = c16 && h->precision == RU_PREC_BF16X3 && C == 16 && (h->fusion & RU_FUSE_GN_BWD_APPLY);
    // form of the gradients dy2 / dy1 that enter the two data-gradient convs (and the two weight gradients): split hi / lo packets -- the direct conv kernel
    // copies them global -> LDS (the Winograd-z routes for them, RU_WZ=2 / 3 of round 5, were measured slower twice and are retired: profiles/r05_notes.txt)
    const bool ds16 = c16;
    const int dgrad_products = c16 ? 1 : h->grad_products();      // (for the kernel choice / partial count only: a split-form input never takes the Winograd-z kernel)
    float *coef2 = nullptr, *coef1 = nullptr;
    int rc = gn_bwd(h, A, s, sv.y2, dout, sv.g2, P(h, params, bp.n2w), kSlope, dy2, G(h, grads, bp.n2w), G(h, grads, bp.n2b), N, C, V, sums2,
                    fa ? &coef2 : nullptr, ds16);
    if (rc) return rc;
    // the published gradients of the 16-channel level have ONE reader each, the data-gradient conv that follows: where conv3_mx_kernel<GRAD> takes the shape they are
    // written in the gradient-operand form of the MX scheme (bf16 main + e4m3 cross terms, a per-voxel exponent; RU_MXG=0 or one-product gradients: the split form)
    const bool g16 = fa && h->grad_products() == 3 && conv3_mxg_usable(N, C, C, D, H, W);
    const GbApply gb2{sv.y2, dout, &sv.g2, coef2, g16};
    // weight gradients whose dy has no other producer role (no fused apply: dy2 / dy1 are complete when gn_bwd returns) leave the chain:
    // on the side stream they run beside the data-gradient convs and the GroupNorm passes of the chain (a 188-register weight-gradient
    // workgroup leaves a third wave's registers free on its CU: the memory-bound passes fit beside it)
    const bool aside = !A.dry && (h->fusion & RU_FUSE_SIDE_STREAM) && c16 && h->precision == RU_PREC_BF16X3 && !fa && !trace_on();
    hipStream_t sw = s;
    if (aside) { rc = side_fork(h, s); if (rc) return rc; sw = h->side; }
    rc = wgrad3_run(A, sw, h->wgrad_mode(), sv.y1, &sv.g1, dy2, G(h, grads, bp.conv2), N, C, C, D, H, W, c16, c16, nullptr, ds16, fa ? &gb2 : nullptr);
    if (rc) return rc;
    float* da1 = A.alloc((size_t)N * C * V);
    Conv3Args d2{};
    d2.x = dy2; d2.wp = h->pack + bp.pk_d2; d2.y = da1; d2.mode = h->precision; d2.products = h->grad_products(); d2.wfrag = h->fpack + bp.fk_d2; d2.in_c16 = c16; d2.out_c16 = c16; d2.in_s16 = ds16; d2.in_g16 = g16; d2.N = N; d2.Cin = C; d2.Cout = C; d2.D = D; d2.H = H; d2.W = W;
    // the data-gradient conv of conv2 takes the GroupNorm-backward sums of norm1 in its epilogue (its output IS the gradient w.r.t.
    // LeakyReLU(norm1(y1))): no separate reduce pass over (y1, da1)
    const bool fuse1 = c16 && h->precision == RU_PREC_BF16X3 && conv3_sb_bst_usable(N, C, D, H, W) && (h->fusion & RU_FUSE_GN_BWD_STATS);
    FusedSums sums1;
    if (fuse1) {
        sums1.nblk = conv3_sb_tiles_per_sample(N, C, C, D, H, W, dgrad_products);
        sums1.part = A.alloc((size_t)N * C * sums1.nblk * 2);
        sums1.coef = A.alloc((size_t)N * C * 3);
        d2.bst_y = sv.y1; d2.bst_k = sv.g1.k; d2.bst_slope = kSlope; d2.stat_partials = sums1.part;  // constants written by the forward finalize
        if (!A.dry && h->tails()) {                      // ... and the conv's last workgroup finalizes them (coefficients, dgamma, dbeta of norm1)
            fill_bwd_tail(h, d2.fin, sums1, sv.g1, P(h, params, bp.n1w), G(h, grads, bp.n1w), G(h, grads, bp.n1b), N, C, V, 1);
            sums1.done = true;
        }
    }
    t_hint_inst = c16 ? (C == 16 ? INST_CONV16_DGRAD : INST_CONV_DEEP_DGRAD) : -1;
    RU_RUN(conv3_launch(d2, s));
    float* dy1 = A.alloc((size_t)N * C * V);
    rc = gn_bwd(h, A, s, sv.y1, da1, sv.g1, P(h, params, bp.n1w), kSlope, dy1, G(h, grads, bp.n1w), G(h, grads, bp.n1b), N, C, V, fuse1 ? &sums1 : nullptr,
                fa ? &coef1 : nullptr, ds16);
    if (rc) return rc;
    const GbApply gb1{sv.y1, da1, &sv.g1, coef1, g16};
    if (aside) { rc = side_fork(h, s); if (rc) return rc; }
    rc = wgrad3_run(A, sw, h->wgrad_mode(), sv.x, sv.xg, dy1, G(h, grads, bp.conv1), N, C, C, D, H, W, c16, c16, nullptr, ds16, fa ? &gb1 : nullptr);
    if (rc) return rc;
    float* dx = A.alloc((size_t)N * C * V);
    Conv3Args d1{};
    d1.x = dy1; d1.wp = h->pack + bp.pk_d1; d1.y = dx; d1.add = dout; d1.mode = h->precision; d1.products = h->grad_products(); d1.wfrag = h->fpack + bp.fk_d1;       // skip path: dx = dout + dgrad(conv1)
    d1.in_c16 = c16; d1.out_c16 = c16; d1.in_s16 = ds16; d1.in_g16 = g16;
    d1.N = N; d1.Cin = C; d1.Cout = C; d1.D = D; d1.H = H; d1.W = W;
    if (nx) nx->out = FusedSums();
    if (nx && fuse1 && bp.down < 0) {                   // same shape and kernel choice as d2: dx = dout + dgrad(conv1) IS the gradient entering nx
        nx->out.nblk = conv3_sb_tiles_per_sample(N, C, C, D, H, W, dgrad_products);
        nx->out.part = A.alloc((size_t)N * C * nx->out.nblk * 2);
        nx->out.coef = A.alloc((size_t)N * C * 3);
        d1.bst_y = nx->y; d1.bst_k = nx->k; d1.bst_slope = nx->slope; d1.stat_partials = nx->out.part;
        if (!A.dry && h->tails() && nx->g) {
            fill_bwd_tail(h, d1.fin, nx->out, *nx->g, nx->gamma, nx->dgamma, nx->dbeta, N, C, V, 1);
            nx->out.done = true;
        }
    }
    t_hint_inst = c16 ? (C == 16 ? INST_CONV16_DGRAD : INST_CONV_DEEP_DGRAD) : -1;
    RU_RUN(conv3_launch(d1, s));
    if (bp.down < 0) { *dxprev_out = dx; return RU_OK; }
    // down-sampling conv backward (Appendix A2): 1x1 over the space-to-depth view
    const int Cp = bp.cin_down;
    {
        Wgrad1Args w{};
        w.x = c16 ? sv.xprev : sv.xs2d; w.dy = dx; w.dw = G(h, grads, bp.down); w.ldw = 8 * Cp; w.c16 = c16; w.tap_split = c16 ? Cp : 0; w.defer = red_for(s);
        if (c16) { w.s2d = 1; w.Dc = D; w.Hc = H; w.Wc = W; }
        w.ws_bytes = wgrad1_workspace_bytes(N, 8 * Cp, C, V);
        w.ws = A.alloc(w.ws_bytes / sizeof(float));
        w.N = N; w.Cin = 8 * Cp; w.Cout = C; w.V = V;
        RU_RUN(wgrad1_launch(w, s));
    }
    float* t = c16 ? nullptr : A.alloc((size_t)N * 8 * Cp * V);
    Conv1Args c1{};
    c1.x0 = dx; c1.C0 = C; c1.y = t; c1.out_slope = 1.f; c1.N = N; c1.Cout = 8 * Cp; c1.V = V;
    float* dxp = A.alloc((size_t)N * Cp * V * 8);
    if (c16) {                                                   // transposed conv scattered straight into the fine tensor
        c1.wT = h->pack + bp.pk_down16 + (size_t)8 * Cp * C; c1.ldw = C;          // [tap*Cp + c][C]
        c1.y = dxp; c1.s2d = 2; c1.Dc = D; c1.Hc = H; c1.Wc = W;
        c1.add = join;                                           // the skip gradient joins in the store
        if (joined) *joined = true;                              // (decided by structure, not by the pointer: the dry walk has null pointers)
        if (nx && (h->fusion & RU_FUSE_GN_BWD_STATS) && conv1_16_bst_nblk(c1) > 0) {
            // the scattered gradient (+ skip gradient) IS what enters nx (norm2 of the last block one level up): its GroupNorm-backward sums
            // are taken in this store pass -- no reduce pass over (y, d) at the finer level
            nx->out.nblk = conv1_16_bst_nblk(c1);
            nx->out.part = A.alloc((size_t)N * Cp * nx->out.nblk * 2);
            nx->out.coef = A.alloc((size_t)N * Cp * 3);
            c1.bst_y = nx->y; c1.bst_k = nx->k; c1.bst_slope = nx->slope; c1.stat_partials = nx->out.part;
        }
        RU_RUN(conv1_16_launch(c1, s));
    } else {
        c1.wT = P(h, params, bp.down); c1.ldw = 8 * Cp;
        RU_RUN(conv1_launch(c1, s));
        RU_RUN(d2s_launch(t, dxp, N, Cp, 2 * D, 2 * H, 2 * W, s));
    }
    *dxprev_out = dxp;
    return RU_OK;
}

// crit: the incoming gradient is the criterion's (ru_unet_backward_criterion): formed inside the head's first pass where that pass exists
// (4-channel head path), else materialised into the workspace first
static int unet_backward_impl(ru_unet* h, const float* params, const float* dprobs, float* grads, float* dx_in, Arena& A, hipStream_t s,
                              const CritGradArgs* crit = nullptr) {
    const int N = h->N, depth = h->depth;
    std::vector<int> Dl(depth), Hl(depth), Wl(depth);
    for (int i = 0; i < depth; ++i) { Dl[i] = h->D >> i; Hl[i] = h->H >> i; Wl[i] = h->W >> i; }
    auto Vl = [&](int i) { return (size_t)Dl[i] * Hl[i] * Wl[i]; };
    const int C0 = h->ch[0];
    t_hint_c = C0;
    RU_RUN(fill_launch(grads, 0.f, h->total, s));     // dead parameters keep zero gradient
    // head
    const bool c16 = h->c16;
    const bool head4 = c16 && conv3_sb4_usable(N, h->nout, C0, Dl[0], Hl[0], Wl[0]);      // few channels: one 4-channel copy feeds both head kernels
    float* d4 = head4 ? A.alloc((size_t)N * 4 * Vl(0)) : nullptr;
    // the 4-channel copy feeds the head's weight gradient only in the Cin <= 16 && Cout <= 16 branch of wgrad3_run: a wider first level
    // takes the generic kernel, which reads the class gradient in NCDHW form -- it is then written as well
    const bool need_dlog = !head4 || C0 > 16;
    float* dlog = need_dlog ? A.alloc((size_t)N * h->nout * Vl(0)) : nullptr;
    const size_t wsb = bias_grad_workspace_bytes(N, h->nout, Vl(0));
    float* wsp = A.alloc(wsb / sizeof(float) + 1);
    const bool crit_fused = crit && head4 && !need_dlog;
    float* dpb = (!(head4 && !need_dlog) && (crit || A.dry)) ? A.alloc((size_t)N * h->nout * Vl(0)) : nullptr;      // (the dry walk sizes for either entry point)
    if (crit && !crit_fused) {                                   // no pass to ride on: the criterion's gradient is written out like a caller would
        RU_RUN(crit_grad_launch(h->probs, crit->target, crit->sums, crit->count, crit->w_dice, crit->w_bce, crit->bgw, crit->priority, dpb, N, h->nout, Vl(0), s));
        dprobs = dpb;
    }
    if (crit_fused) {                                            // criterion gradient, sigmoid backward, 4-channel copy and bias gradient in one pass
        RU_RUN(head_grad_c4_crit_launch(h->probs, *crit, d4, G(h, grads, h->conv_out_b), N, h->nout, Vl(0), wsp, wsb, s));
    } else if (head4) {                                          // sigmoid backward, 4-channel copy and bias gradient in one pass
        RU_RUN(head_grad_c4_launch(h->probs, dprobs, d4, G(h, grads, h->conv_out_b), N, h->nout, Vl(0), wsp, wsb, s));
        if (need_dlog) RU_RUN(sigmoid_bwd_launch(h->probs, dprobs, dlog, (size_t)N * h->nout * Vl(0), s));
    } else {
        RU_RUN(sigmoid_bwd_launch(h->probs, dprobs, dlog, (size_t)N * h->nout * Vl(0), s));
        RU_RUN(bias_grad_launch(dlog, G(h, grads, h->conv_out_b), N, h->nout, Vl(0), wsp, wsb, s));
    }
    int rc = wgrad3_run(A, s, h->wgrad_mode(), h->head_in, nullptr, dlog, G(h, grads, h->conv_out_w), N, C0, h->nout, Dl[0], Hl[0], Wl[0], c16, false,
                        C0 <= 16 ? d4 : nullptr);
    if (rc) return rc;
    float* dcur_buf = A.alloc((size_t)N * C0 * Vl(0));
    Conv3Args dh{};
    dh.x = dlog; dh.wp = h->pack + h->pk_out_d; dh.y = dcur_buf; dh.mode = h->precision; dh.wfrag = h->fpack + h->fk_out_d; dh.out_c16 = c16; dh.N = N; dh.Cin = h->nout; dh.Cout = C0; dh.D = Dl[0]; dh.H = Hl[0]; dh.W = Wl[0];
    if (head4) {                                                 // few input channels: tap-pair kernel on the 4-channel copy
        float* wf4 = A.alloc(conv3_sb4_frag_bytes(C0) / sizeof(float) + 64);
        RU_RUN(conv3_sb4_pack_weights(P(h, params, h->conv_out_w), wf4, C0, h->nout, 1, s));
        dh.x = d4; dh.wfrag = wf4; dh.in_c4 = 1;
    }
    // the head's data gradient is the gradient entering norm2 of the last decoder block: its GroupNorm-backward sums are taken here
    const bool no_bst = !(h->fusion & RU_FUSE_GN_BWD_STATS);
    const BlockSave* hb = (depth >= 2 && !h->dec_s[0].empty()) ? &h->dec_s[0].back() : nullptr;
    FusedSums hsums;
    if (head4 && hb && !no_bst && conv3_sb_bst_usable(N, C0, Dl[0], Hl[0], Wl[0])) {
        hsums.nblk = conv3_sb_tiles_per_sample(N, h->nout, C0, Dl[0], Hl[0], Wl[0]);
        hsums.part = A.alloc((size_t)N * C0 * hsums.nblk * 2);
        hsums.coef = A.alloc((size_t)N * C0 * 3);
        dh.bst_y = hb->y2; dh.bst_k = hb->g2.k; dh.bst_slope = kSlope; dh.stat_partials = hsums.part;
        if (!A.dry && h->tails()) {
            const BlockP& lb = *hb->bp;
            fill_bwd_tail(h, dh.fin, hsums, hb->g2, P(h, params, lb.n2w), G(h, grads, lb.n2w), G(h, grads, lb.n2b), N, C0, Vl(0), 1);
            hsums.done = true;
        }
    }
    RU_RUN(conv3_launch(dh, s));
    const float* dcur = dcur_buf;
    std::vector<const float*> dskip(depth - 1, nullptr);
    FusedSums carry = hsums;                                     // sums of the norm2 the current gradient enters, taken by the kernel that produced it
    // decoder stages, reverse of execution order (forward ran i = depth-2 .. 0)
    for (int i = 0; i <= depth - 2; ++i) {
        for (int j = (int)h->dec_s[i].size() - 1; j >= 0; --j) {
            const bool fused2 = carry.nblk > 0 && j == (int)h->dec_s[i].size() - 1;
            const FusedSums sin = carry;
            rc = block_bwd(h, params, grads, A, s, h->dec_s[i][j], dcur, &dcur, nullptr, nullptr, fused2 ? &sin : nullptr);
            if (rc) return rc;
        }
        carry = FusedSums();
        const DecSave& ds = h->dstage[i];
        const int Ci = h->ch[i], Cc = h->ch[i + 1];
        const size_t V = Vl(i);
        // decoder_convs1x1[i] over cat([skip, v]) (model.py:424-425)
        const float* wdec = P(h, params, h->dec1_w[i]);          // [Ci][2Ci]
        float* gdec = G(h, grads, h->dec1_w[i]);
        float* dsk = A.alloc((size_t)N * Ci * V);
        float* dv = A.alloc((size_t)N * Ci * V);
        float* dpre = A.alloc((size_t)N * Ci * V);
        // Ci <= 32: the weight-gradient kernel of the concat 1x1 also forms its DATA gradient from the dy tile it has staged (both halves,
        // LeakyReLU backward of the up-sampled half from the staged v): one pass over dcur / skip / v instead of two
        const bool dg_fused = c16 && Ci <= 32 && (h->fusion & RU_FUSE_PW_DGRAD);
        if (c16) {                                               // one pass over dcur for both halves of the (never materialised) concat
            Wgrad1Args w{};
            w.x = ds.skip; w.x1 = ds.v; w.C0 = Ci; w.dy = dcur; w.dw = gdec; w.ldw = 2 * Ci; w.c16 = 1; w.defer = red_for(s);
            w.ws_bytes = wgrad1_workspace_bytes(N, 2 * Ci, Ci, V);
            w.ws = A.alloc(w.ws_bytes / sizeof(float));
            w.N = N; w.Cin = 2 * Ci; w.Cout = Ci; w.V = V;
            if (dg_fused) { w.dg_w = wdec; w.dg_ldw = 2 * Ci; w.dg_y0 = dsk; w.dg_y1 = dpre; w.dg_mask_slope = kSlope; }
            RU_RUN(wgrad1_launch(w, s));
        } else {
            rc = wgrad1_run(A, s, ds.skip, dcur, gdec, 2 * Ci, N, Ci, Ci, V, c16);
            if (rc) return rc;
            rc = wgrad1_run(A, s, ds.v, dcur, A.dry ? nullptr : gdec + Ci, 2 * Ci, N, Ci, Ci, V, c16);
            if (rc) return rc;
        }
        Conv1Args a1{};
        a1.x0 = dcur; a1.C0 = Ci; a1.y = dsk; a1.out_slope = 1.f; a1.N = N; a1.Cout = Ci; a1.V = V;
        Conv1Args a2 = a1;
        a2.y = dv;
        if (dg_fused) {
            // (written by the weight-gradient launch above)
        } else if (c16) {                                               // [out][in] = the transposed pack [2Ci][Ci]: rows 0..Ci-1 skip half, Ci.. up half
            a1.wT = h->pack + h->pk_decT[i]; a1.ldw = Ci;
            a2.wT = h->pack + h->pk_decT[i] + (size_t)Ci * Ci; a2.ldw = Ci;
            a2.y = dpre; a2.mask = ds.v; a2.mask_slope = kSlope; // LeakyReLU backward (model.py:422) fused into the store
            Conv1Args a12 = a1;                                  // both halves in one pass over dcur: [2Ci][Ci] weights, split output
            a12.Cout = 2 * Ci; a12.Cout0 = Ci; a12.y1 = dpre; a12.mask = ds.v; a12.mask_slope = kSlope;
            RU_RUN(conv1_16_launch(a12, s));
        } else {
            a1.wT = wdec; a1.ldw = 2 * Ci;
            a2.wT = A.dry ? nullptr : wdec + Ci; a2.ldw = 2 * Ci;
            RU_RUN(conv1_launch(a1, s));
            RU_RUN(conv1_launch(a2, s));
        }
        dskip[i] = dsk;
        // LeakyReLU backward from the output v (model.py:422; Appendix A4), then upsampling[i][1] (1x1) and Trilinear
        if (!c16) RU_RUN(lrelu_bwd_launch(ds.v, dv, dpre, (size_t)N * Ci * V, kSlope, s));
        Conv1Args a3{};
        a3.x0 = dpre; a3.C0 = Ci; a3.out_slope = 1.f; a3.N = N; a3.Cout = Cc; a3.V = V;
        float* dz = A.alloc((size_t)N * Cc * Vl(i + 1));
        float* du = nullptr;
        if (c16) {                                               // adjoint of up(conv(z)): everything after the transpose-interpolation is coarse
            float* dzc = A.alloc((size_t)N * Ci * Vl(i + 1));
            RU_RUN(up2_bwd16_launch(dpre, dzc, N, Ci, Dl[i + 1], Hl[i + 1], Wl[i + 1], s));
            rc = wgrad1_run(A, s, ds.z, dzc, G(h, grads, h->up_w[i]), Cc, N, Cc, Ci, Vl(i + 1), true);
            if (rc) return rc;
            a3.x0 = dzc; a3.y = dz; a3.V = Vl(i + 1);
            a3.wT = h->pack + h->pk_upT[i]; a3.ldw = Ci;          // [Cc][Ci] = [out][in]
            // dz is the gradient entering norm2 of the block that produced z (the last block of the next decoder stage, or of the deepest
            // encoder level): its GroupNorm-backward sums are taken in this launch's store pass
            carry = FusedSums();
            // (a middle decoder stage with zero blocks: z is then a 1x1 output, not a block's -- no norm2 whose sums could ride here; the
            // deepest encoder level's y2 is a tensor of ANOTHER level and must not be read as bst_y)
            const BlockSave* zb = (i + 1 <= depth - 2) ? (!h->dec_s[i + 1].empty() ? &h->dec_s[i + 1].back() : nullptr)
                                  : (!h->enc_s[depth - 2].empty() ? &h->enc_s[depth - 2].back() : nullptr);
            if (zb && (zb->g2.k || A.dry) && (h->fusion & RU_FUSE_GN_BWD_STATS) && conv1_16_bst_nblk(a3) > 0) {
                carry.nblk = conv1_16_bst_nblk(a3);
                carry.part = A.alloc((size_t)N * Cc * carry.nblk * 2);
                carry.coef = A.alloc((size_t)N * Cc * 3);
                a3.bst_y = zb->y2; a3.bst_k = zb->g2.k; a3.bst_slope = kSlope; a3.stat_partials = carry.part;
            }
            RU_RUN(conv1_16_launch(a3, s));
        } else {
            rc = wgrad1_run(A, s, ds.u, dpre, G(h, grads, h->up_w[i]), Cc, N, Cc, Ci, V, false);
            if (rc) return rc;
            du = A.alloc((size_t)N * Cc * V);
            a3.y = du;
            a3.wT = P(h, params, h->up_w[i]); a3.ldw = Cc;
            RU_RUN(conv1_launch(a3, s));
            RU_RUN(up2_bwd_launch(du, dz, N, Cc, Dl[i + 1], Hl[i + 1], Wl[i + 1], s));
        }
        dcur = dz;
    }
    // encoder levels, deepest first; the skip gradient joins at each level's input
    for (int i = depth - 2; i >= 0; --i) {
        bool joined = false;
        FusedSums snext = carry;                                 // sums of the NEXT block's norm2, taken by the kernel that produced its incoming gradient
        carry = FusedSums();
        for (int j = (int)h->enc_s[i].size() - 1; j >= 0; --j) {
            GNNext nx{};
            nx.slope = kSlope;
            // the gradient this block produces enters norm2 of the block before it -- for the level's first (down-sampling) block that is the
            // last block one level up (its stride-2 transpose scatters into that level and adds the skip gradient there)
            const BlockSave* pb = j >= 1 ? &h->enc_s[i][j - 1] : (i >= 1 ? (h->enc_s[i - 1].empty() ? nullptr : &h->enc_s[i - 1].back())
                                                                        : (h->first_s.empty() ? nullptr : &h->first_s.back()));
            if (pb && pb->g2.k) {
                nx.y = pb->y2; nx.k = pb->g2.k; nx.g = &pb->g2;
                nx.gamma = P(h, params, pb->bp->n2w); nx.dgamma = G(h, grads, pb->bp->n2w); nx.dbeta = G(h, grads, pb->bp->n2b);
            }
            const FusedSums sin = snext;
            rc = block_bwd(h, params, grads, A, s, h->enc_s[i][j], dcur, &dcur, j == 0 ? dskip[i] : nullptr, j == 0 ? &joined : nullptr, sin.nblk > 0 ? &sin : nullptr,
                           (pb && (nx.y || A.dry)) ? &nx : nullptr);
            if (rc) return rc;
            snext = nx.out;
        }
        carry = snext;                                           // (from the level's down-sampling block: the sums of the finer level's last norm2)
        if (!joined) {
            float* sum = A.alloc((size_t)N * h->ch[i] * Vl(i));
            RU_RUN(add_launch(dcur, dskip[i], sum, (size_t)N * h->ch[i] * Vl(i), s));
            dcur = sum;
            carry = FusedSums();                                 // (the sums were those of the gradient before the skip gradient joined)
        }
    }
    FusedSums sfirst = carry;
    for (int j = (int)h->first_s.size() - 1; j >= 0; --j) {
        // the gradient a first-level block produces enters norm2 of the block before it, or (j = 0) norm_input (no activation: slope 1)
        GNNext nx{};
        nx.slope = j >= 1 ? kSlope : 1.0f;
        if (j >= 1) {
            const BlockSave& pb = h->first_s[j - 1];
            nx.y = pb.y2; nx.k = pb.g2.k; nx.g = &pb.g2;
            nx.gamma = P(h, params, pb.bp->n2w); nx.dgamma = G(h, grads, pb.bp->n2w); nx.dbeta = G(h, grads, pb.bp->n2b);
        } else {
            nx.y = h->y0; nx.k = h->g0.k; nx.g = &h->g0;
            nx.gamma = P(h, params, h->nin_w); nx.dgamma = G(h, grads, h->nin_w); nx.dbeta = G(h, grads, h->nin_b);
        }
        const FusedSums sin = sfirst;
        rc = block_bwd(h, params, grads, A, s, h->first_s[j], dcur, &dcur, nullptr, nullptr, sin.nblk > 0 ? &sin : nullptr, (nx.k || A.dry) ? &nx : nullptr);
        if (rc) return rc;
        sfirst = nx.out;
    }
    // norm_input (no activation: slope 1) and conv_input
    t_hint_c = C0;
    float* dy0 = A.alloc((size_t)N * C0 * Vl(0));
    // voxel-major engine, no d/d(input) wanted: the gradient w.r.t. the stem conv's output is consumed by the stem's weight gradient alone, so
    // the GroupNorm-backward apply of norm_input is computed in that kernel's staging (wgrad3_tz<1,1,3>) and never written
    // (only the Cin <= 16 && Cout <= 16 branch of wgrad3_run computes the apply while staging: a wider stem takes the generic
    // weight-gradient kernel, which reads a WRITTEN dy0)
    const bool fuse0 = c16 && h->precision == RU_PREC_BF16X3 && (h->fusion & RU_FUSE_GN_BWD_APPLY) && !dx_in && h->x_in4_planned && C0 <= 16 && kInCh <= 4;
    float* coef0 = nullptr;
    // a stem wider than 16 channels: its weight gradient is the generic mixed-layout kernel (x NCDHW, dy voxel-major float32), which does
    // not read the split form
    const bool split0 = c16 && C0 <= 16;
    rc = gn_bwd(h, A, s, h->y0, dcur, h->g0, P(h, params, h->nin_w), 1.0f, dy0, G(h, grads, h->nin_w), G(h, grads, h->nin_b), N, C0, Vl(0), sfirst.nblk > 0 ? &sfirst : nullptr,
                fuse0 ? &coef0 : nullptr, split0);
    if (rc) return rc;
    const GbApply gb0{h->y0, dcur, &h->g0, coef0};
    rc = wgrad3_run(A, s, h->wgrad_mode(), h->x_in, nullptr, dy0, G(h, grads, h->conv_in), N, kInCh, C0, Dl[0], Hl[0], Wl[0], false, c16, C0 <= 16 ? h->x_in4 : nullptr, split0,
                    fuse0 ? &gb0 : nullptr);
    if (rc) return rc;
    if (dx_in) {
        // d/d(input): not needed by training (train.py:201-210), offered for gradient checks
        float* wpd = A.alloc(conv3_packed_floats(C0, kInCh));
        float* wfd = A.alloc(conv3_sb_frag_bytes(C0, kInCh) / sizeof(float) + 64);
        Conv3Args di{};
        if (c16) {                                               // dy0 is voxel-major: the split-bf16 kernel reads it
            RU_RUN(conv3_sb_pack_weights(P(h, params, h->conv_in), wfd, kInCh, C0, 1, s));
            di.mode = RU_PREC_BF16X3; di.wfrag = wfd; di.in_c16 = 1; di.in_s16 = split0 ? 1 : 0;
        } else {
            RU_RUN(conv3_pack_weights(P(h, params, h->conv_in), wpd, kInCh, C0, 1, s));
        }
        di.x = dy0; di.wp = wpd; di.y = dx_in; di.N = N; di.Cin = C0; di.Cout = kInCh; di.D = Dl[0]; di.H = Hl[0]; di.W = Wl[0];
        RU_RUN(conv3_launch(di, s));
    }
    if (t_red && !t_red->side.e.empty()) RU_RUN(wgrad_reduce_flush(t_red->side, h->side));   // (stream order on the side stream: behind its weight-gradient kernels)
    if (!A.dry) { rc = side_join(h, s); if (rc) return rc; }     // the caller's stream sees every gradient
    if (t_red && !t_red->main.e.empty()) RU_RUN(wgrad_reduce_flush(t_red->main, s));         // every other queued partial-sum reduction, one launch
    return RU_OK;
}

static int check_dims(ru_unet* h, int N, int D, int H, int W) {
    const int m = 1 << (h->depth - 1);
    RU_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "ru_unet: bad extents");
    RU_REQUIRE(D % m == 0 && H % m == 0 && W % m == 0, "ru_unet: D,H,W must be divisible by %d (model.py:361 stride-2 convs)", m);
    return RU_OK;
}

extern "C" size_t ru_unet_workspace_bytes(ru_unet_t h, int N, int D, int H, int W, int training) {
    if (!h || check_dims(h, N, D, H, W) != RU_OK) return 0;
    ru_unet tmp = *h;            // dry walk on a copy: does not disturb a live forward state
    tmp.probe_ev.clear();        // (the copy must not own the handle's HIP events: its destructor would destroy them)
    tmp.probe_on = false;
    tmp.sink = nullptr;
    tmp.side = nullptr;          // (nor its side stream / events / tickets)
    tmp.tickets = nullptr;
    tmp.fork_ev.clear();
    tmp.N = N; tmp.D = D; tmp.H = H; tmp.W = W; tmp.training = training != 0;
    Arena A;
    A.dry = true;
    if (unet_forward_impl(&tmp, nullptr, nullptr, nullptr, A, nullptr) != RU_OK) return 0;
    if (training) {
        if (unet_backward_impl(&tmp, nullptr, nullptr, nullptr, (float*)1, A, nullptr) != RU_OK) return 0;
    }
    return A.need() + 4096;
}

extern "C" int ru_unet_forward(ru_unet_t h, const float* params, const float* x, float* probs, int N, int D, int H, int W, int training,
                               void* ws, size_t ws_bytes, ru_stream_t stream) {
    RU_REQUIRE(h && params && x && probs && ws, "ru_unet_forward: null argument");
    int rc = check_dims(h, N, D, H, W);
    if (rc) return rc;
    h->have_fwd = false;
    h->N = N; h->D = D; h->H = H; h->W = W; h->training = training != 0;
    h->ws = (char*)ws; h->ws_bytes = ws_bytes;
    Arena A;
    A.dry = false; A.base = (char*)ws; A.cap = ws_bytes;
    if (!h->tickets && (h->fusion & RU_FUSE_TAIL_FINALIZE)) {    // only the RU_FUSION_ON devtools route gets here (ru_unet_create / set_fusion may run without a device):
        hipError_t e = hipMalloc((void**)&h->tickets, 256 * sizeof(unsigned));      // the bit is on from the first (warm-up) call; ru_unet_set_fusion allocates eagerly
        if (e != hipSuccess) { h->tickets = nullptr; return hip_fail(e, "ru_unet_forward: ticket words"); }
    }
    if (h->tickets && (h->fusion & RU_FUSE_TAIL_FINALIZE)) {
        // every step starts from zeroed tickets (a memset node under capture): a launch that faulted or was aborted after taking a ticket cannot
        // make a later launch that draws the same word skip its finalize
        hipError_t e = hipMemsetAsync(h->tickets, 0, 256 * sizeof(unsigned), (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "ru_unet_forward: ticket words");
    }
    h->ticket_next = 0;
    ru::t_sink = h->probe_families ? h->sink : nullptr;
    rc = unet_forward_impl(h, params, x, probs, A, (hipStream_t)stream);
    ru::t_sink = nullptr;
    if (rc) return rc;
    if (A.failed) { set_error("ru_unet_forward: workspace too small (%zu bytes given)", ws_bytes); return RU_ENOMEM; }
    h->fwd_end = A.off;
    h->fwd_keep = A.keep;
    h->have_fwd = true;
    return RU_OK;
}

static int backward_entry(ru_unet_t h, const float* params, const float* dprobs, const CritGradArgs* crit, float* grads, float* dx, ru_stream_t stream) {
    if (!h->have_fwd || !h->training) { set_error("ru_unet_backward: needs a preceding training-mode ru_unet_forward"); return RU_ESTATE; }
    Arena A;
    A.dry = false; A.base = h->ws; A.cap = h->ws_bytes; A.off = h->fwd_end; A.keep = h->fwd_keep;
    ru::t_sink = h->probe_families ? h->sink : nullptr;
    h->red.main.e.clear();
    h->red.side.e.clear();
    ru::t_red = (h->fusion & RU_FUSE_BATCH_WREDUCE) && !trace_on() ? &h->red : nullptr;
    // a training forward packs only the fragment forms its switches launch (sb_pack_forms): a switch flipped between that forward and this backward would make
    // a launch read fragments that were never packed -- refused instead (round-5 advisor finding; tools and tests toggle between steps, which is fine)
    RU_REQUIRE(h->precision != RU_PREC_BF16X3 || h->pack_sig == conv3_sb_switch_signature(),
               "ru_unet_backward: RU_WZ / RU_MX / RU_MXG changed since the forward whose packs this backward reads (signature %d then, %d now)", h->pack_sig, conv3_sb_switch_signature());
    int rc = unet_backward_impl(h, params, dprobs, grads, dx, A, (hipStream_t)stream, crit);
    ru::t_red = nullptr;
    ru::t_sink = nullptr;
    if (rc && h->side) {                                 // an error return must not leave the side stream running behind the caller's back
        (void)side_join(h, (hipStream_t)stream);
        h->fork_used = 0;
    }
    if (rc) return rc;
    if (A.failed) { set_error("ru_unet_backward: workspace too small"); return RU_ENOMEM; }
    return RU_OK;
}
extern "C" int ru_unet_backward_criterion(ru_unet_t h, const float* params, const float* target, const double* sums, double count,
                                          float w_dice, float w_bce, float bg_weight, float priority, float* grads, float* dx, ru_stream_t stream) {
    RU_REQUIRE(h && params && target && sums && grads && count > 0.0, "ru_unet_backward_criterion: null argument");
    const CritGradArgs cg{target, sums, count, w_dice, w_bce, bg_weight, priority};
    return backward_entry(h, params, nullptr, &cg, grads, dx, stream);
}

extern "C" int ru_unet_backward(ru_unet_t h, const float* params, const float* dprobs, float* grads, float* dx, ru_stream_t stream) {
    RU_REQUIRE(h && params && dprobs && grads, "ru_unet_backward: null argument");
    return backward_entry(h, params, dprobs, nullptr, grads, dx, stream);
}

extern "C" int ru_unet_gn_stats(ru_unet_t h, int idx, float* mean, float* rstd, ru_stream_t stream) {
    RU_REQUIRE(h && h->have_fwd, "ru_unet_gn_stats: no forward state");
    if (idx < 0) return (int)h->gn_order.size();
    RU_REQUIRE(idx < (int)h->gn_order.size() && mean && rstd, "ru_unet_gn_stats: bad index");
    const size_t bytes = (size_t)h->N * kGroups * sizeof(float);
    hipError_t e = hipMemcpyAsync(mean, h->gn_order[idx].mean, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rstd, h->gn_order[idx].rstd, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "ru_unet_gn_stats");
    return RU_OK;
}

// ====================================================================== op-level C-ABI
extern "C" const char* ru_last_error(void) { return g_err; }
extern "C" int ru_version(void) { return 100; }
extern "C" int ru_device_ok(void) {
    int n = 0;
    return (hipGetDeviceCount(&n) == hipSuccess && n > 0) ? 1 : 0;
}

struct WsCarver {
    char* base; size_t cap, off = 0; bool failed = false;
    WsCarver(void* b, size_t c) : base((char*)b), cap(c) {}
    float* take(size_t nfloats) {
        const size_t bytes = align_up(nfloats * sizeof(float), 256);
        if (!base || off + bytes > cap) { failed = true; return nullptr; }
        float* p = (float*)(base + off);
        off += bytes;
        return p;
    }
};
#define RU_WS_OK(w) do { if ((w).failed) { set_error("workspace too small"); return RU_ENOMEM; } } while (0)

extern "C" size_t ru_conv3d_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W, int k) {
    size_t b = 4096;
    if (k == 3) {
        b += align_up(conv3_packed_floats(Cin, Cout) * 4, 256) + align_up(conv3_packed_floats(Cout, Cin) * 4, 256);
        b += align_up(conv3_sb_frag_bytes(Cin, Cout), 256) + align_up(conv3_sb_frag_bytes(Cout, Cin), 256);
        b += align_up(wgrad3_workspace_bytes(N, Cin, Cout, D, H, W), 256) + align_up(bias_grad_workspace_bytes(N, Cout, (size_t)D * H * W), 256);
    } else if (k == 1) {
        b += align_up((size_t)Cin * Cout * 4, 256) + align_up(wgrad1_workspace_bytes(N, Cin, Cout, (size_t)D * H * W), 256);
    } else if (k == 2) {
        const size_t Vo = (size_t)(D / 2) * (H / 2) * (W / 2);
        b += align_up((size_t)N * 8 * Cin * Vo * 4, 256) + align_up((size_t)8 * Cin * Cout * 4, 256) + align_up(wgrad1_workspace_bytes(N, 8 * Cin, Cout, Vo), 256);
    }
    return b;
}

extern "C" int ru_conv3d_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout, int D, int H, int W, int k,
                             void* ws, size_t ws_bytes, ru_stream_t stream) {
    return ru_conv3d_fwd_p(x, w, bias, y, N, Cin, Cout, D, H, W, k, RU_PREC_F32, ws, ws_bytes, stream);
}

// pack one 3x3x3 weight for `precision` into the workspace and fill the weight fields of `a`
static int prep_conv3_weights(Conv3Args& a, const float* w, int Cin_f, int Cout_f, int mode, int precision, int W, WsCarver& C, hipStream_t s) {
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    a.mode = precision;
    if (conv3_effective_mode(precision, W) == RU_PREC_BF16X3) {
        void* wf = C.take(conv3_sb_frag_bytes(cin_conv, cout_conv) / 4 + 64);
        RU_WS_OK(C);
        a.wfrag = wf;
        return conv3_sb_pack_weights(w, wf, Cin_f, Cout_f, mode, s);
    }
    float* wp = C.take(conv3_packed_floats(cin_conv, cout_conv));
    RU_WS_OK(C);
    a.wp = wp;
    return conv3_pack_weights(w, wp, Cin_f, Cout_f, mode, s);
}

extern "C" int ru_conv3d_fwd_p(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout, int D, int H, int W, int k,
                               int precision, void* ws, size_t ws_bytes, ru_stream_t stream) {
    RU_REQUIRE(x && w && y, "ru_conv3d_fwd: null argument");
    RU_REQUIRE(precision == RU_PREC_F32 || precision == RU_PREC_BF16X3, "ru_conv3d_fwd: bad precision");
    hipStream_t s = (hipStream_t)stream;
    WsCarver C(ws, ws_bytes);
    if (k == 3) {
        Conv3Args a{};
        int rc = prep_conv3_weights(a, w, Cin, Cout, 0, precision, W, C, s);
        if (rc) return rc;
        a.x = x; a.bias = bias; a.y = y; a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
        return conv3_launch(a, s);
    }
    RU_REQUIRE(!bias, "ru_conv3d_fwd: bias only supported for k=3 (model.py:348)");
    if (k == 1) {
        float* wT = C.take((size_t)Cin * Cout);
        RU_WS_OK(C);
        int rc = transpose_launch(w, wT, Cout, Cin, s);
        if (rc) return rc;
        Conv1Args a{};
        a.x0 = x; a.C0 = Cin; a.wT = wT; a.ldw = Cout; a.y = y; a.out_slope = 1.f; a.N = N; a.Cout = Cout; a.V = (size_t)D * H * W;
        return conv1_launch(a, s);
    }
    if (k == 2) {
        RU_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "ru_conv3d_fwd: k=2 needs even extents");
        const size_t Vo = (size_t)(D / 2) * (H / 2) * (W / 2);
        float* xs = C.take((size_t)N * 8 * Cin * Vo);
        float* wT = C.take((size_t)8 * Cin * Cout);
        RU_WS_OK(C);
        int rc = s2d_launch(x, xs, N, Cin, D, H, W, s);
        if (rc) return rc;
        rc = transpose_launch(w, wT, Cout, 8 * Cin, s);
        if (rc) return rc;
        Conv1Args a{};
        a.x0 = xs; a.C0 = 8 * Cin; a.wT = wT; a.ldw = Cout; a.y = y; a.out_slope = 1.f; a.N = N; a.Cout = Cout; a.V = Vo;
        return conv1_launch(a, s);
    }
    set_error("ru_conv3d_fwd: unsupported kernel size %d", k);
    return RU_EINVAL;
}

extern "C" int ru_conv3d_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int D, int H, int W, int k,
                                  void* ws, size_t ws_bytes, ru_stream_t stream) {
    return ru_conv3d_bwd_data_p(dy, w, dx, N, Cin, Cout, D, H, W, k, RU_PREC_F32, ws, ws_bytes, stream);
}

extern "C" int ru_conv3d_bwd_data_p(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int D, int H, int W, int k,
                                    int precision, void* ws, size_t ws_bytes, ru_stream_t stream) {
    RU_REQUIRE(dy && w && dx, "ru_conv3d_bwd_data: null argument");
    RU_REQUIRE(precision == RU_PREC_F32 || precision == RU_PREC_BF16X3, "ru_conv3d_bwd_data: bad precision");
    hipStream_t s = (hipStream_t)stream;
    WsCarver C(ws, ws_bytes);
    if (k == 3) {
        Conv3Args a{};
        int rc = prep_conv3_weights(a, w, Cin, Cout, 1, precision, W, C, s);
        if (rc) return rc;
        a.x = dy; a.y = dx; a.N = N; a.Cin = Cout; a.Cout = Cin; a.D = D; a.H = H; a.W = W;
        return conv3_launch(a, s);
    }
    if (k == 1) {
        Conv1Args a{};
        a.x0 = dy; a.C0 = Cout; a.wT = w; a.ldw = Cin; a.y = dx; a.out_slope = 1.f; a.N = N; a.Cout = Cin; a.V = (size_t)D * H * W;
        return conv1_launch(a, s);
    }
    if (k == 2) {
        RU_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "ru_conv3d_bwd_data: k=2 needs even extents");
        const size_t Vo = (size_t)(D / 2) * (H / 2) * (W / 2);
        float* t = C.take((size_t)N * 8 * Cin * Vo);
        RU_WS_OK(C);
        Conv1Args a{};
        a.x0 = dy; a.C0 = Cout; a.wT = w; a.ldw = 8 * Cin; a.y = t; a.out_slope = 1.f; a.N = N; a.Cout = 8 * Cin; a.V = Vo;
        int rc = conv1_launch(a, s);
        if (rc) return rc;
        return d2s_launch(t, dx, N, Cin, D, H, W, s);
    }
    set_error("ru_conv3d_bwd_data: unsupported kernel size %d", k);
    return RU_EINVAL;
}

extern "C" int ru_conv3d_bwd_weight(const float* x, const float* dy, float* dw, float* db, int N, int Cin, int Cout, int D, int H, int W, int k,
                                    void* ws, size_t ws_bytes, ru_stream_t stream) {
    return ru_conv3d_bwd_weight_p(x, dy, dw, db, N, Cin, Cout, D, H, W, k, RU_PREC_F32, ws, ws_bytes, stream);
}

extern "C" int ru_conv3d_bwd_weight_p(const float* x, const float* dy, float* dw, float* db, int N, int Cin, int Cout, int D, int H, int W, int k,
                                      int precision, void* ws, size_t ws_bytes, ru_stream_t stream) {
    RU_REQUIRE(x && dy && dw, "ru_conv3d_bwd_weight: null argument");
    RU_REQUIRE(precision == RU_PREC_F32 || precision == RU_PREC_BF16X3, "ru_conv3d_bwd_weight: bad precision");
    hipStream_t s = (hipStream_t)stream;
    WsCarver C(ws, ws_bytes);
    if (k == 3) {
        Wgrad3Args a{};
        a.x = x; a.dy = dy; a.dw = dw; a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.mode = precision;
        a.ws_bytes = wgrad3_workspace_bytes(N, Cin, Cout, D, H, W);
        a.ws = C.take(a.ws_bytes / 4);
        RU_WS_OK(C);
        int rc = wgrad3_launch(a, s);
        if (rc) return rc;
        if (db) {
            const size_t b = bias_grad_workspace_bytes(N, Cout, (size_t)D * H * W);
            float* p = C.take(b / 4 + 1);
            RU_WS_OK(C);
            return bias_grad_launch(dy, db, N, Cout, (size_t)D * H * W, p, b, s);
        }
        return RU_OK;
    }
    RU_REQUIRE(!db, "ru_conv3d_bwd_weight: bias only supported for k=3");
    if (k == 1) {
        Wgrad1Args a{};
        a.x = x; a.dy = dy; a.dw = dw; a.ldw = Cin; a.N = N; a.Cin = Cin; a.Cout = Cout; a.V = (size_t)D * H * W;
        a.ws_bytes = wgrad1_workspace_bytes(N, Cin, Cout, a.V);
        a.ws = C.take(a.ws_bytes / 4);
        RU_WS_OK(C);
        return wgrad1_launch(a, s);
    }
    if (k == 2) {
        RU_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "ru_conv3d_bwd_weight: k=2 needs even extents");
        const size_t Vo = (size_t)(D / 2) * (H / 2) * (W / 2);
        float* xs = C.take((size_t)N * 8 * Cin * Vo);
        RU_WS_OK(C);
        int rc = s2d_launch(x, xs, N, Cin, D, H, W, s);
        if (rc) return rc;
        Wgrad1Args a{};
        a.x = xs; a.dy = dy; a.dw = dw; a.ldw = 8 * Cin; a.N = N; a.Cin = 8 * Cin; a.Cout = Cout; a.V = Vo;
        a.ws_bytes = wgrad1_workspace_bytes(N, 8 * Cin, Cout, Vo);
        a.ws = C.take(a.ws_bytes / 4);
        RU_WS_OK(C);
        return wgrad1_launch(a, s);
    }
    set_error("ru_conv3d_bwd_weight: unsupported kernel size %d", k);
    return RU_EINVAL;
}

extern "C" size_t ru_groupnorm_workspace_bytes(int N, int C, size_t V) {
    return 4096 + align_up((size_t)N * C * gn_stats_tiles(V) * 2 * 4, 256) + 5 * align_up((size_t)N * C * 4, 256) + align_up((size_t)N * C * 3 * 4, 256);
}

extern "C" int ru_groupnorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* mean, float* rstd,
                                int N, int C, size_t V, int G, float eps, float slope, void* ws, size_t ws_bytes, ru_stream_t stream) {
    RU_REQUIRE(x && gamma && beta && y && mean && rstd, "ru_groupnorm_fwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    WsCarver Cw(ws, ws_bytes);
    const int nblk = gn_stats_tiles(V);
    float* part = Cw.take((size_t)N * C * nblk * 2);
    float* scale = Cw.take((size_t)N * C);
    float* shift = Cw.take((size_t)N * C);
    RU_WS_OK(Cw);
    int rc = gn_stats_launch(x, part, N, C, V, s);
    if (rc) return rc;
    rc = gn_finalize_launch(part, nblk, gamma, beta, mean, rstd, scale, shift, N, C, V, G, eps, s);
    if (rc) return rc;
    return gn_apply_launch(x, scale, shift, residual, y, N, C, V, slope, s);
}

extern "C" int ru_groupnorm_bwd(const float* x, const float* gamma, const float* beta, const float* mean, const float* rstd, const float* dy,
                                float* dx, float* dgamma, float* dbeta, int N, int C, size_t V, int G, float slope,
                                void* ws, size_t ws_bytes, ru_stream_t stream) {
    RU_REQUIRE(x && gamma && beta && mean && rstd && dy && dx, "ru_groupnorm_bwd: null argument");
    RU_REQUIRE(C % G == 0, "ru_groupnorm_bwd: C %% G != 0");
    hipStream_t s = (hipStream_t)stream;
    WsCarver Cw(ws, ws_bytes);
    const int nblk = gn_bwd_tiles(V);
    float* part = Cw.take((size_t)N * C * nblk * 2);
    float* scale = Cw.take((size_t)N * C);
    float* shift = Cw.take((size_t)N * C);
    float* coef = Cw.take((size_t)N * C * 3);
    RU_WS_OK(Cw);
    hipLaunchKernelGGL(gn_scale_shift_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, s, gamma, beta, mean, rstd, scale, shift, N, C, G);
    RU_CHECK_LAUNCH("gn_scale_shift_kernel");
    int rc = gn_bwd_reduce_launch(x, dy, scale, shift, mean, rstd, slope, part, N, C, V, G, s);
    if (rc) return rc;
    rc = gn_bwd_finalize_launch(part, nblk, gamma, mean, rstd, coef, dgamma, dbeta, N, C, V, G, s);
    if (rc) return rc;
    return gn_bwd_apply_launch(x, dy, scale, shift, coef, slope, dx, N, C, V, s);
}

extern "C" int ru_leaky_relu_fwd(const float* x, float* y, size_t n, float slope, ru_stream_t stream) { return lrelu_fwd_launch(x, y, n, slope, (hipStream_t)stream); }
extern "C" int ru_leaky_relu_bwd(const float* y, const float* dy, float* dx, size_t n, float slope, ru_stream_t stream) { return lrelu_bwd_launch(y, dy, dx, n, slope, (hipStream_t)stream); }
extern "C" int ru_upsample2x_trilinear_fwd(const float* x, float* y, int N, int C, int D, int H, int W, ru_stream_t stream) { return up2_fwd_launch(x, y, N, C, D, H, W, (hipStream_t)stream); }
extern "C" int ru_upsample2x_trilinear_bwd(const float* dy, float* dx, int N, int C, int D, int H, int W, ru_stream_t stream) { return up2_bwd_launch(dy, dx, N, C, D, H, W, (hipStream_t)stream); }
extern "C" int ru_sigmoid_fwd(const float* x, float* y, size_t n, ru_stream_t stream) { return sigmoid_launch(x, y, n, (hipStream_t)stream); }

extern "C" size_t ru_criterion_workspace_bytes(int N, int C, size_t V) { return 4096 + (size_t)N * C * crit_tiles(V) * 3 * sizeof(float); }
extern "C" int ru_criterion_sums(const float* p, const float* g, double* sums, int N, int C, size_t V, float bg_weight, void* ws, size_t ws_bytes, ru_stream_t stream) {
    RU_REQUIRE(p && g && sums, "ru_criterion_sums: null argument");
    return crit_sums_launch(p, g, sums, N, C, V, bg_weight, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int ru_criterion_grad(const float* p, const float* g, const double* sums, double count, float w_dice, float w_bce, float bg_weight,
                                 float priority, float* dp, int N, int C, size_t V, ru_stream_t stream) {
    RU_REQUIRE(p && g && sums && dp && count > 0, "ru_criterion_grad: bad argument");
    return crit_grad_launch(p, g, sums, count, w_dice, w_bce, bg_weight, priority, dp, N, C, V, (hipStream_t)stream);
}
extern "C" int ru_criterion_value(const double* sums_host, int C, double count, double priority, double* dice, double* bce) {
    RU_REQUIRE(sums_host && C > 0 && count > 0, "ru_criterion_value: bad argument");
    double acc = 0.0;
    for (int c = 0; c < C; ++c) acc += 2.0 * (sums_host[c] + 1e-6) / (sums_host[C + c] + 2e-6);   // loss.py:114-117
    if (dice) *dice = priority * (1.0 - acc / C);                                                  // loss.py:122
    if (bce) *bce = -sums_host[2 * C] / count;                                                     // loss.py:79
    return RU_OK;
}
extern "C" int ru_criterion_value_device(const double* sums, int C, double count, double priority, double w_dice, double w_bce, double* out3,
                                         ru_stream_t stream) {
    RU_REQUIRE(sums && out3 && C > 0 && C <= 64 && count > 0, "ru_criterion_value_device: bad argument");
    return crit_value_launch(sums, C, count, priority, w_dice, w_bce, out3, (hipStream_t)stream);
}
extern "C" int ru_adam_amsgrad_step(float* w, const float* g, float* m, float* v, float* vmax, size_t n, float lr, float beta1, float beta2,
                                    float eps, float weight_decay, int step, ru_stream_t stream) {
    RU_REQUIRE(w && g && m && v && vmax, "ru_adam_amsgrad_step: null argument");
    return adam_launch(w, g, m, v, vmax, n, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream);
}

extern "C" int ru_adam_step(float* w, const float* g, float* m, float* v, float* vmax_or_null, size_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, ru_stream_t stream) {
    RU_REQUIRE(w && g && m && v, "ru_adam_step: null argument");
    return adam_launch(w, g, m, v, vmax_or_null, n, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream);
}

// ---------------------------------------------------------------- inference post-processing
extern "C" int ru_tta_merge(const float* probs, int K, unsigned flips, float* mean_out, unsigned char* mask, unsigned long long* counts,
                            int C, int D, int H, int W, ru_stream_t stream) {
    RU_REQUIRE(probs && mask && counts, "ru_tta_merge: null argument");
    return tta_merge_launch(probs, K, flips, mean_out, mask, counts, C, D, H, W, (hipStream_t)stream);
}
extern "C" int ru_compose_labels(const unsigned char* mask, const unsigned long long* counts, unsigned long long et_min, unsigned char* labels,
                                 size_t V, ru_stream_t stream) {
    RU_REQUIRE(mask && counts && labels, "ru_compose_labels: null argument");
    return compose_labels_launch(mask, counts, et_min, labels, V, (hipStream_t)stream);
}

extern "C" int ru_dice_accumulate(const unsigned long long* counts, double* acc, int N, int C, int nacc, ru_stream_t stream) {
    return dice_accumulate_launch(counts, acc, N, C, nacc, (hipStream_t)stream);
}
extern "C" int ru_dice_counts(const float* p, const float* g, unsigned long long* counts, int N, int C, size_t V, ru_stream_t stream) {
    RU_REQUIRE(p && g && counts && N > 0 && C > 0, "ru_dice_counts: bad argument");
    return dice_counts_launch(p, g, counts, N * C, V, (hipStream_t)stream);
}

// ====================================================================== C16 layout hooks (tests / probes)
extern "C" int ru_layout_convert(const float* src, float* dst, int N, int C, size_t V, int to_c16, ru_stream_t stream) {
    RU_REQUIRE(src && dst && N > 0, "ru_layout_convert: bad argument");
    return layout_convert_launch(src, dst, N, C, V, to_c16, (hipStream_t)stream);
}

extern "C" int ru_upsample2x_trilinear_fwd_l(const float* x, float* y, int N, int C, int D, int H, int W, float out_slope, ru_stream_t stream) {
    RU_REQUIRE(x && y && N > 0 && C > 0 && D > 0 && H > 0 && W > 0, "ru_upsample2x_trilinear_fwd_l: bad argument");
    return up2_fwd16_launch(x, y, N, C, D, H, W, out_slope, (hipStream_t)stream);
}
extern "C" int ru_upsample2x_trilinear_bwd_l(const float* dy, float* dx, int N, int C, int D, int H, int W, ru_stream_t stream) {
    RU_REQUIRE(dy && dx && N > 0 && C > 0 && D > 0 && H > 0 && W > 0, "ru_upsample2x_trilinear_bwd_l: bad argument");
    return up2_bwd16_launch(dy, dx, N, C, D, H, W, (hipStream_t)stream);
}
extern "C" int ru_conv3d_fwd_l(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout, int D, int H, int W,
                               int flags, void* ws, size_t ws_bytes, ru_stream_t stream) {
    RU_REQUIRE(x && w && y, "ru_conv3d_fwd_l: null argument");
    hipStream_t s = (hipStream_t)stream;
    WsCarver C(ws, ws_bytes);
    Conv3Args a{};
    void* wf = C.take(conv3_sb_frag_bytes(Cin, Cout) / 4 + 64);
    RU_WS_OK(C);
    const bool f32 = (flags & 16) != 0;                          // exact-f32 arithmetic on voxel-major tensors (conv3_f32c_kernel)
    a.products = (flags & 32) ? 2 : 0;                           // the input is an activation tensor: fp16 + MX-fp8 products where the shape has that kernel (conv3_mx.hpp)
    RU_REQUIRE(!f32 || ((flags & 3) != 0 && !(flags & (4 | 8))), "ru_conv3d_fwd_l: the exact-f32 form needs a voxel-major side and takes neither the 4-channel copy nor a split-form input");
    // flag bit 6: x (float32, voxel-major) is a GRADIENT -- where conv3_mx_kernel<GRAD> takes the shape it is converted to the gradient-operand form of the MX scheme
    // (conv3_mxg_split_launch; in a training step wgrad3_tz<1,0,3,3> writes that form) and convolved with bf16 main + MX cross products; elsewhere the bit is ignored
    const bool gop = (flags & 64) && !f32 && (flags & 3) == 3 && !(flags & (4 | 8)) && !bias && conv3_mxg_usable(N, Cin, Cout, D, H, W);
    int rc = f32 ? conv3_f32c_pack_weights(w, wf, Cin, Cout, 0, s) : conv3_sb_pack_weights(w, wf, Cin, Cout, 0, s, gop);
    if (rc) return rc;
    a.mode = f32 ? RU_PREC_F32 : RU_PREC_BF16X3; a.wfrag = wf;
    a.in_c16 = flags & 1; a.out_c16 = (flags >> 1) & 1;
    a.in_s16 = (flags >> 3) & 1;                                 // x is voxel-major in SPLIT form (hi / lo bf16 packets, as gn_bwd_apply16 publishes it)
    RU_REQUIRE(!a.in_s16 || a.in_c16, "ru_conv3d_fwd_l: the split form is a voxel-major layout");
    a.x = x; a.bias = bias; a.y = y; a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    if (gop) {
        const size_t nvox = (size_t)N * D * H * W;
        float* g16 = C.take(nvox * 16);
        RU_WS_OK(C);
        rc = conv3_mxg_split_launch(x, g16, nvox, s);
        if (rc) return rc;
        a.x = g16; a.in_s16 = 1; a.in_g16 = 1; a.products = 0;
    }
    if (flags & 4) {                                             // x NCDHW with Cin <= 4: 4-channel copy + tap-pair kernel
        RU_REQUIRE(!(flags & 1) && conv3_sb4_usable(N, Cin, Cout, D, H, W), "ru_conv3d_fwd_l: shape does not fit the 4-channel kernel");
        const size_t V = (size_t)D * H * W;
        float* x4 = C.take((size_t)N * 4 * V);
        void* wf4 = C.take(conv3_sb4_frag_bytes(Cout) / 4 + 64);
        RU_WS_OK(C);
        rc = pad_to_c4_launch(x, x4, N, Cin, V, s);
        if (rc) return rc;
        rc = conv3_sb4_pack_weights(w, wf4, Cin, Cout, 0, s);
        if (rc) return rc;
        a.x = x4; a.wfrag = wf4; a.in_c4 = 1;
    }
    return f32 ? conv3_launch(a, s) : conv3_sb_launch(a, s);
}

extern "C" int ru_conv3d_bwd_weight_l(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int D, int H, int W,
                                      int flags, void* ws, size_t ws_bytes, ru_stream_t stream) {
    RU_REQUIRE(x && dy && dw, "ru_conv3d_bwd_weight_l: null argument");
    WsCarver C(ws, ws_bytes);
    Wgrad3Args a{};
    a.x = x; a.dy = dy; a.dw = dw; a.mode = RU_PREC_BF16X3; a.x_c16 = flags & 1; a.dy_c16 = (flags >> 1) & 1;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    a.ws_bytes = wgrad3_workspace_bytes(N, Cin, Cout, D, H, W);
    a.ws = C.take(a.ws_bytes / 4);
    RU_WS_OK(C);
    return wgrad3_launch(a, (hipStream_t)stream);
}

// ====================================================================== training input pipeline (dataloader.py)
extern "C" size_t ru_zscore_workspace_bytes(int C, size_t V) { return zscore_workspace_bytes(C, V) + 256; }
extern "C" int ru_zscore_stats(const float* image, double* stats, int C, size_t V, void* ws, size_t ws_bytes, ru_stream_t stream) {
    RU_REQUIRE(image && stats && C > 0 && V > 0, "ru_zscore_stats: bad argument");
    return zscore_stats_launch(image, stats, C, V, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int ru_augment_patch(const float* image, const unsigned char* label, const float* mean, const float* inv_std, int C, int D, int H, int W,
                                const int* crop_lo, const int* patch, const double* scale, int flags, const float* gain, const float* bias,
                                float* data_out, float* target_out, ru_stream_t stream) {
    RU_REQUIRE(image && label && mean && inv_std && crop_lo && patch && scale && gain && bias && data_out && target_out, "ru_augment_patch: null argument");
    RU_REQUIRE(C > 0 && C <= RU_AUG_MAXC, "ru_augment_patch: 1..%d channels", RU_AUG_MAXC);
    const int dims[3] = {D, H, W};
    AugmentArgs a{};
    a.image = image; a.label = label; a.data = data_out; a.target = target_out; a.C = C; a.D = D; a.H = H; a.W = W; a.flags = flags;
    for (int i = 0; i < 3; ++i) {
        RU_REQUIRE(patch[i] > 0 && crop_lo[i] >= 0 && crop_lo[i] + patch[i] <= dims[i], "ru_augment_patch: the crop must lie inside the volume");
        RU_REQUIRE(scale[i] > 0.0, "ru_augment_patch: scale must be positive");
        a.lo[i] = crop_lo[i]; a.P[i] = patch[i]; a.scale[i] = scale[i];
    }
    for (int c = 0; c < C; ++c) { a.mean[c] = mean[c]; a.istd[c] = inv_std[c]; a.gain[c] = gain[c]; a.bias[c] = bias[c]; }
    return augment_patch_launch(a, (hipStream_t)stream);
}
