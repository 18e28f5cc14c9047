// C ABI of libnnb.so (see include/nnb.h): workspace carving, weight loading and the
// host side of the lock-step search / self-play loops.  No device memory is
// allocated here: the engine lives in the caller's workspace.
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/nnb.h"
#include "c4_bits.h"
#include "engine_layout.h"
#include "kernels.h"

using namespace nnb;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t err__ = (expr);                                                             \
        if (err__ != hipSuccess)                                                               \
            return fail(NNB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

constexpr size_t kAlign = 256;
constexpr int kMaxSims = 9000;           // first_child is 16 bit: 1 + 7*sims <= 65535
constexpr size_t kFusedLdsLimit = 160 * 1024;   // LDS of one CU
// The fused tree kernel's share of it when a search's whole pool would take more: sixteen games per CU (four waves per
// SIMD, 4096 games resident at once on 256 CUs), the pool's first nodes in LDS and the rest in memory.  The kernel
// is a chain of dependent loads per game, so its throughput follows the games in flight: 800 simulations under PUCT
// with rollouts, 4096 games: 32 M sims/s with the whole pool in LDS (112 KB, one game per CU), 55 / 76 / 95 / 120 /
// 145 / 187 M with a share of 78 / 52 / 39 / 26 / 13 / 10 KB, and no more below that (tools/tree_occupancy.py).
#ifndef NNB_FUSED_LDS_KB
#define NNB_FUSED_LDS_KB 10
#endif
constexpr size_t kFusedLdsShare = (size_t)NNB_FUSED_LDS_KB * 1024;
constexpr int kHalvesMinSlots = 1024;    // below this a round is not split over two streams
constexpr int kStepGamesHost = 16;       // games per block of the tree step kernel (halves are cut on a block boundary)
constexpr int kProfClasses = 7;
constexpr int kProfSamples = 16;         // event pairs per class per drain

struct Carver {
    unsigned char *base;
    size_t off = 0;
    explicit Carver(void *b) : base(static_cast<unsigned char *>(b)) {}
    template <typename T> T *take(size_t count) {
        off = (off + kAlign - 1) / kAlign * kAlign;
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel on a device, not to an engine: one high-water
// mark per device for the whole process, only ever raised (engines and nnb_search_static share the kernel).
constexpr int kMaxDevices = 64;
size_t g_fused_lds[kMaxDevices] = {};
std::mutex g_fused_lds_mutex;

int ensure_tree_fused_lds(size_t lds) {
    int device = -1;
    HIP_TRY(hipGetDevice(&device));
    if (device < 0 || device >= kMaxDevices) return fail(NNB_ERR_ARG, "device ordinal %d out of range", device);
    std::lock_guard<std::mutex> lock(g_fused_lds_mutex);
    if (lds > g_fused_lds[device]) {
        HIP_TRY(prepare_tree_fused(lds));
        g_fused_lds[device] = lds;
    }
    return NNB_OK;
}

struct ProfClass {
    hipEvent_t start[kProfSamples], stop[kProfSamples];
    int used = 0;
    double total_ms = 0.0;
    long long launches = 0;
    long long seen = 0;
};

}  // namespace

struct nnb_engine {
    nnb_config cfg;
    EngineDev dev;
    NetBuffers buf;
    std::vector<NetWeights> weights;
    std::vector<int> weights_loaded;
    int32_t *host_counters = nullptr;      // pinned, 16 ints
    unsigned long long *host_stats = nullptr;   // pinned, 8 u64
    hipStream_t side_stream = nullptr;     // second forward pass of a two-network arena, second chain of a self-play round
    hipStream_t side_checked_for = nullptr; // the caller's stream that side_stream was last seen to run beside
    bool side_checked = false;
    int side_replaced = 0;                  // candidate second streams discarded so far
    hipEvent_t fork_event = nullptr, join_event = nullptr;
    // nnb_selfplay_stream: games stay seated between calls
    bool stream_on = false;
    int stream_slots = 0;
    long long stream_steps = 0;            // lock-step rounds since the restart (their parity selects the leaf buffers)
    long long stream_finished = 0;         // games finished since the restart, as of the last call's end
    long long stream_prev[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // cumulative statistics as of the last call's end
    nnb_search_params stream_params = {};
    RoundDev stream_round = {};
    bool policy_heads = true;              // the search in progress reads the policy heads (PUCT), not only the value
    bool shared_chip = false;              // two forward passes run side by side on two streams
    int row_bound = 0;                     // > 0: no leaf batch of the rounds being launched holds more rows than this
                                           //      (the occupied slots at the host's last look; they only go down until
                                           //      the next look), so the forward pass may pick its small-batch kernels
    bool prof_on = false;
    bool prof_events = false;
    int prof_stride = 1;
    ProfClass prof[kProfClasses];
};

namespace {

// Search state of G concurrent games / roots whose per-game node pools in device memory hold `cap`
// nodes (the fused tree kernel keeps its pool in LDS and only hands the root and its children over,
// so 8 suffice for it), with a log table for up to max_sims simulations.
void carve_state(size_t G, size_t cap, int max_sims, int cache_slots, Carver &c, EngineDev &d) {
    d.max_games = (int32_t)G; d.pool_cap = (int32_t)cap; d.max_sims = max_sims; d.n_games = 0;
    d.nodes = c.take<Node>(G * cap);
    d.prior = c.take<float>(G * cap);
    d.log_table = c.take<double>(max_sims + 8);
    d.root_boards = c.take<u64>(2 * G);
    d.root_player = c.take<int8_t>(G);
    d.root_side = c.take<int8_t>(G);
    d.active = c.take<int32_t>(G);
    d.n_nodes = c.take<int32_t>(G);
    d.sims_done = c.take<int32_t>(G);
    d.pending = c.take<int32_t>(G);
    d.path_len = c.take<int32_t>(G);
    d.path = c.take<uint16_t>(G * kPathSlots);
    d.ply = c.take<int32_t>(G);
    d.game_stats = c.take<int32_t>(8 * G);
    d.traj_index = c.take<int32_t>(G);
    d.game_id = c.take<int32_t>(G);
    d.move_index = c.take<int32_t>(G);
    d.root_noise = c.take<float>(7 * G);
    d.leaf_side = c.take<u64>(4 * G);
    d.leaf_legal = c.take<uint8_t>(4 * G);
    d.leaf_count = c.take<int32_t>(64);
    d.leaf_value = c.take<float>(2 * G);
    d.leaf_logp = c.take<float>(14 * G);
    d.leaf_prior = c.take<float>(14 * G);
    d.counters = c.take<int32_t>(64);
    d.stats64 = c.take<unsigned long long>(32);
    d.cache_gen = c.take<uint32_t>(G);
    d.pend_key = c.take<u64>(2 * G);
    d.cache_slots = cache_slots;
    d.cache = cache_slots > 0 ? c.take<CacheEntry>(G * (size_t)cache_slots) : nullptr;
}

// Returns the size of the leading part of the workspace that must start out zeroed (search state);
// the activation and weight areas behind it are fully overwritten before they are read.
size_t carve(const nnb_config &cfg, Carver &c, EngineDev &d, NetBuffers &b, std::vector<NetWeights> &w) {
    const size_t G = cfg.max_games;
    carve_state(G, 1 + 7 * (size_t)cfg.max_sims, cfg.max_sims, cfg.eval_cache_slots, c, d);
    // ticket and flag words of the split linear layer's block pairs (they only ever grow; zero at the start)
    b.fc_pairs = fc_sync_pairs(cfg.max_games);
    b.fcsync = b.fcsync_b1 = nullptr;
    if (cfg.net_precision == NNB_NET_F32S) {
        b.fcsync = c.take<uint32_t>(2 * (size_t)b.fc_pairs);
        b.fcsync_b1 = c.take<uint32_t>(2 * (size_t)b.fc_pairs);
    }
    // maxima of the four-blocks-per-board convolutions of small batches (zero between passes, so zero at the start)
    b.spread_max = b.spread_max_b1 = nullptr;
    b.spread_f32 = b.spread_f32_b1 = nullptr;
    if (cfg.net_precision == NNB_NET_F32S) {
        b.spread_max = c.take<uint32_t>(2 * kConvSpreadMaxBoards);
        b.spread_max_b1 = c.take<uint32_t>(2 * kConvSpreadMaxBoards);
    }
    b.sync_err = d.counters ? d.counters + kSyncErr : nullptr;
    b.clk = c.take<unsigned long long>(8);              // profiling stamps; kernels get the pointer only while profiling is on
    b.fcq = b.fcq_b1 = b.fchand = b.fchand_b1 = nullptr;
    b.heads_all = 1; b.write_hidden = 0;
    c.take<unsigned char>(kAlign);
    const size_t state_bytes = c.off;
    if (cfg.net_precision == NNB_NET_F32) {
        b.act1 = c.take<float>(G * 56 * 128);
        b.act2 = c.take<float>(G * 72 * 128);
        b.act3 = c.take<float>(G * 90 * 128);
        b.fc = c.take<float>(G * 1024);
        b.fc_b1 = c.take<float>(G * 1024);
        b.fcpart = b.fcpart_b1 = nullptr;
        const bool two = true;     // the second leaf batch: the other network of an arena, or the other half of the slots
        b.act1_b1 = two ? c.take<float>(G * 56 * 128) : nullptr;
        b.act2_b1 = two ? c.take<float>(G * 72 * 128) : nullptr;
        b.act3_b1 = two ? c.take<float>(G * 90 * 128) : nullptr;
    } else if (cfg.net_precision == NNB_NET_F32S) {      // hi + lo halves: 4 bytes per value, plus a scale per board
        const bool two = true;
        b.act1 = b.act1_b1 = nullptr;
        b.fc = c.take<float>(G * 1024);                 // hidden activations, written for nnb_debug_activations only
        b.fc_b1 = nullptr;
        b.act2 = c.take<float>(G * 72 * 128);
        b.act3 = c.take<float>(G * 90 * 128);
        b.act2_inv = c.take<float>(G);
        b.act3_inv = c.take<float>(G);
        b.fcpart = b.fcpart_b1 = nullptr;
        b.fchand = c.take<float>((G + 128) * 1024);     // a row block may reach 127 rows past the last row
        b.fchand_b1 = c.take<float>((G + 128) * 1024);
        b.fcq = c.take<float>(G * 256);
        b.fcq_b1 = c.take<float>(G * 256);
        const size_t spread_boards = G < (size_t)kConvSpreadMaxBoards ? G : (size_t)kConvSpreadMaxBoards;
        b.spread_f32 = c.take<float>(spread_boards * kConvSpreadFloatsPerBoard);
        b.spread_f32_b1 = c.take<float>(spread_boards * kConvSpreadFloatsPerBoard);
        b.act2_b1 = two ? c.take<float>(G * 72 * 128) : nullptr;
        b.act3_b1 = two ? c.take<float>(G * 90 * 128) : nullptr;
        b.act2_inv_b1 = two ? c.take<float>(G) : nullptr;
        b.act3_inv_b1 = two ? c.take<float>(G) : nullptr;
    } else {                                // half activations; the 2->128 layer's output stays in LDS
        b.act1 = nullptr;
        b.act2 = reinterpret_cast<float *>(c.take<uint16_t>(G * 72 * 128));
        b.act3 = reinterpret_cast<float *>(c.take<uint16_t>(G * 90 * 128));
        b.fc = b.fc_b1 = nullptr;
        b.fcpart = c.take<float>(2 * G * 1024);
        b.fcpart_b1 = c.take<float>(2 * G * 1024);
        const bool two = true;
        b.act1_b1 = nullptr;
        b.act2_b1 = two ? reinterpret_cast<float *>(c.take<uint16_t>(G * 72 * 128)) : nullptr;
        b.act3_b1 = two ? reinterpret_cast<float *>(c.take<uint16_t>(G * 90 * 128)) : nullptr;
    }
    b.max_rows = cfg.max_games;
    w.resize(cfg.n_weight_sets);
    for (auto &s : w) {
        s.conv0_w = c.take<float>(128 * 32);  s.conv0_b = c.take<float>(128);
        const bool f32 = cfg.net_precision == NNB_NET_F32;
        const bool split = cfg.net_precision == NNB_NET_F32S;
        s.conv2_b = c.take<float>(128); s.conv4_b = c.take<float>(128); s.fc_b = c.take<float>(1024);
        s.conv2_w = f32 ? c.take<float>(128 * 128 * 16) : nullptr;
        s.conv4_w = f32 ? c.take<float>(128 * 128 * 16) : nullptr;
        s.fc_w = f32 ? c.take<float>((size_t)1024 * 11520) : nullptr;
        s.conv2_wh = f32 ? nullptr : (void *)c.take<uint16_t>(128 * 128 * 16);
        s.conv4_wh = f32 ? nullptr : (void *)c.take<uint16_t>(128 * 128 * 16);
        s.fc_wh = f32 ? nullptr : (void *)c.take<uint16_t>((size_t)1024 * 11520);
        s.conv2_wl = split ? (void *)c.take<uint16_t>(128 * 128 * 16) : nullptr;
        s.conv4_wl = split ? (void *)c.take<uint16_t>(128 * 128 * 16) : nullptr;
        s.fc_wl = split ? (void *)c.take<uint16_t>((size_t)1024 * 11520) : nullptr;
        s.conv0_wh = split ? (void *)c.take<uint16_t>(128 * 32) : nullptr;
        s.conv0_wl = split ? (void *)c.take<uint16_t>(128 * 32) : nullptr;
        s.sp_scales = split ? c.take<SplitScales>(1) : nullptr;
        s.sp_maxbits = split ? (void *)c.take<uint32_t>(4) : nullptr;
        s.heads_w = c.take<float>(8 * 1024);  s.heads_b = c.take<float>(8);
    }
    c.take<unsigned char>(kAlign);          // tail pad
    return state_bytes;
}

int check_cfg(const nnb_config *cfg) {
    if (!cfg) return fail(NNB_ERR_ARG, "config is null");
    if (cfg->max_games < 1 || cfg->max_games > (1 << 20))
        return fail(NNB_ERR_ARG, "max_games %d out of range [1, 2^20]", cfg->max_games);
    if (cfg->max_sims < 1 || cfg->max_sims > kMaxSims)
        return fail(NNB_ERR_ARG, "max_sims %d out of range [1, %d]", cfg->max_sims, kMaxSims);
    if (cfg->n_weight_sets < 1 || cfg->n_weight_sets > 8)
        return fail(NNB_ERR_ARG, "n_weight_sets %d out of range [1, 8]", cfg->n_weight_sets);
    if (cfg->net_precision != NNB_NET_F16 && cfg->net_precision != NNB_NET_F32 && cfg->net_precision != NNB_NET_F32S)
        return fail(NNB_ERR_ARG, "net_precision %d out of range", cfg->net_precision);
    const int cs = cfg->eval_cache_slots;
    if (cs != 0 && (cs < 0 || cs > 65536 || (cs & (cs - 1)) != 0))
        return fail(NNB_ERR_ARG, "eval_cache_slots %d is neither 0 nor a power of two <= 65536", cs);
    return NNB_OK;
}

// ---- optional per-kernel-class timing with HIP events on the launch stream ----
struct ProfScope {
    nnb_engine *e; int cls; hipStream_t st; int slot = -1;
    ProfScope(nnb_engine *e_, int cls_, hipStream_t st_) : e(e_), cls(cls_), st(st_) {
        if (!e->prof_on) return;
        ProfClass &p = e->prof[cls];
        // uniform sample over the launches of one move, so the mean is unbiased
        bool take = p.used < kProfSamples && p.seen % e->prof_stride == 0;
        ++p.seen;
        if (take) { slot = p.used++; (void)hipEventRecord(p.start[slot], st); }
    }
    ~ProfScope() {
        if (slot >= 0) (void)hipEventRecord(e->prof[cls].stop[slot], st);
    }
};

void prof_drain(nnb_engine *e) {          // call only when the stream is idle
    if (!e->prof_on) return;
    for (auto &p : e->prof) {
        for (int i = 0; i < p.used; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.start[i], p.stop[i]) == hipSuccess) {
                p.total_ms += ms;
                ++p.launches;
            }
        }
        p.used = 0;
        p.seen = 0;
    }
}

SearchDev to_dev(const nnb_search_params &p) {
    SearchDev s;
    s.mode = p.mode; s.evaluator = p.evaluator; s.sims = p.sims; s.flags = p.flags; s.rng_move_index = p.rng_move_index; s.rng_game_base = p.rng_game_base;
    s.wset[0] = p.weight_set; s.wset[1] = p.weight_set_colour2_plus1 > 0 ? p.weight_set_colour2_plus1 - 1 : p.weight_set;
    s.side[0] = p.side_of_colour[0]; s.side[1] = p.side_of_colour[1];
    s.c_puct = p.c_puct; s.dirichlet_alpha = p.dirichlet_alpha; s.dirichlet_eps = p.dirichlet_eps;
    s.seed = p.seed;
    return s;
}

int check_params(nnb_engine *e, const nnb_search_params *p, int n) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    if (!p) return fail(NNB_ERR_ARG, "params is null");
    e->row_bound = 0;
    e->shared_chip = false;
    if (n < 1 || n > e->cfg.max_games)
        return fail(NNB_ERR_ARG, "n = %d out of range [1, max_games = %d]", n, e->cfg.max_games);
    if (p->sims < 1 || p->sims > e->cfg.max_sims)
        return fail(NNB_ERR_ARG, "sims = %d out of range [1, max_sims = %d]", p->sims, e->cfg.max_sims);
    if (p->mode != NNB_MODE_COMPAT && p->mode != NNB_MODE_AZ) return fail(NNB_ERR_ARG, "unknown mode %d", p->mode);
    if (p->evaluator < NNB_EVAL_NET || p->evaluator > NNB_EVAL_ROLLOUT)
        return fail(NNB_ERR_ARG, "unknown evaluator %d", p->evaluator);
    if ((p->flags & NNB_FLAG_EVAL_CACHE) && e->cfg.eval_cache_slots == 0)
        return fail(NNB_ERR_STATE, "NNB_FLAG_EVAL_CACHE needs an engine created with eval_cache_slots > 0");
    if ((p->flags & NNB_FLAG_EVAL_CACHE) && p->evaluator == NNB_EVAL_ROLLOUT)
        return fail(NNB_ERR_ARG, "random rollouts are not a function of the position: no evaluation cache");
    if (p->evaluator == NNB_EVAL_ROLLOUT && (p->flags & NNB_FLAG_LOCKSTEP))
        return fail(NNB_ERR_ARG, "the rollout evaluator runs in the fused kernel only (no NNB_FLAG_LOCKSTEP)");
    if (p->mode == NNB_MODE_AZ) {
        if (!(p->dirichlet_eps >= 0.0 && p->dirichlet_eps <= 1.0)) return fail(NNB_ERR_ARG, "dirichlet_eps must be in [0, 1]");
        if (!(p->dirichlet_alpha >= 0.0) || !std::isfinite(p->dirichlet_alpha))
            return fail(NNB_ERR_ARG, "dirichlet_alpha must be finite and >= 0");
    }
    {
        const int sets[2] = {p->weight_set, p->weight_set_colour2_plus1 > 0 ? p->weight_set_colour2_plus1 - 1 : p->weight_set};
        for (int k = 0; k < 2; ++k) {
            if (sets[k] < 0 || sets[k] >= e->cfg.n_weight_sets)
                return fail(NNB_ERR_ARG, "weight set %d out of range [0, %d)", sets[k], e->cfg.n_weight_sets);
            if (p->evaluator == NNB_EVAL_NET && !e->weights_loaded[sets[k]])
                return fail(NNB_ERR_STATE, "weight set %d has not been loaded (nnb_load_weights)", sets[k]);
        }
    }
    if (!(p->c_puct >= 0.0) || !std::isfinite(p->c_puct)) return fail(NNB_ERR_ARG, "c_puct must be finite and >= 0");
    if (p->round_limit < 0) return fail(NNB_ERR_ARG, "round_limit must be >= 0");
    for (int k = 0; k < 2; ++k)
        if (p->side_of_colour[k] < 0 || p->side_of_colour[k] > 2)
            return fail(NNB_ERR_ARG, "side_of_colour[%d] = %d is not 0 (the mover), 1 or 2", k, p->side_of_colour[k]);
    return NNB_OK;
}

void net_stage(nnb_engine *e, hipStream_t st, int stage, const NetWeights &w, const NetBuffers &b, const u64 *side,
               const uint8_t *legal, float *logp, float *value, float *prior, const int32_t *count_ptr, int n_max) {
    if (e->cfg.net_precision == NNB_NET_F32)
        launch_net_stage(st, stage, w, b, side, legal, logp, value, prior, count_ptr, n_max);
    else if (e->cfg.net_precision == NNB_NET_F32S)
        launch_net_stage_split(st, stage, w, b, side, legal, logp, value, prior, count_ptr, n_max);
    else
        launch_net_stage_f16(st, stage, w, b, side, legal, logp, value, prior, count_ptr, n_max);
}

// Activation buffers of leaf batch `batch`: the second batch of a two-network arena has buffers of
// its own, so the two forward passes can run side by side on two streams.
NetBuffers batch_buffers(const nnb_engine *e, int batch) {
    NetBuffers b = e->buf;
    b.shared_chip = e->shared_chip ? 1 : 0;
    if (!e->prof_on || batch == 1) b.clk = nullptr;      // one writer: launches of the two batches overlap in time
    if (batch == 1) {
        b.fc = b.fc_b1; b.fcpart = b.fcpart_b1;
        b.fcq = b.fcq_b1; b.fchand = b.fchand_b1; b.fcsync = b.fcsync_b1;
        b.act1 = b.act1_b1; b.act2 = b.act2_b1; b.act3 = b.act3_b1;
        b.act2_inv = b.act2_inv_b1; b.act3_inv = b.act3_inv_b1;
        b.spread_f32 = b.spread_f32_b1; b.spread_max = b.spread_max_b1;
    }
    return b;
}

HeadsDev heads_input(const nnb_engine *e, int weight_set, int batch) {
    const NetBuffers b = batch_buffers(e, batch);
    if (e->cfg.net_precision == NNB_NET_F32S) return heads_input_split(e->weights[weight_set], b);
    return e->cfg.net_precision == NNB_NET_F32 ? heads_input_f32(e->weights[weight_set], b)
                                               : heads_input_f16(e->weights[weight_set], b);
}

// Forward pass over one of the engine's compacted leaf batches (count on the device).  With
// `with_heads` false it stops after the linear layer; the tree step computes the heads.
void forward_leaves(nnb_engine *e, hipStream_t st, int weight_set, int batch, int parity, int n_max, bool with_heads) {
    if (e->row_bound > 0 && e->row_bound < n_max) n_max = e->row_bound;
    const int32_t *count_ptr = e->dev.leaf_count + 2 * parity + batch;
    const NetWeights &w = e->weights[weight_set];
    const EngineDev &d = e->dev;
    NetBuffers nb = batch_buffers(e, batch);
    // the reference rule ignores priors: when the tree step finishes the heads itself, the value head is all it reads
    nb.heads_all = (with_heads || e->policy_heads) ? 1 : 0;
    const size_t o = (size_t)batch * d.max_games;
    const bool fused_in = e->cfg.net_precision != NNB_NET_F32;     // conv0 is part of the conv2 launch there
    if (!fused_in) { ProfScope ps(e, 1, st); net_stage(e, st, NET_CONV0, w, nb, d.leaf_side + 2 * o, nullptr, nullptr, nullptr, nullptr, count_ptr, n_max); }
    if (e->cfg.net_precision == NNB_NET_F32S) {
        // all three convolutions in one launch (timed as class 2; class 3 stays empty)
        ProfScope ps(e, 2, st); net_stage(e, st, NET_CONV_PAIR, w, nb, d.leaf_side + 2 * o, nullptr, nullptr, nullptr, nullptr, count_ptr, n_max);
    } else {
        { ProfScope ps(e, 2, st); net_stage(e, st, NET_CONV2, w, nb, d.leaf_side + 2 * o, nullptr, nullptr, nullptr, nullptr, count_ptr, n_max); }
        { ProfScope ps(e, 3, st); net_stage(e, st, NET_CONV4, w, nb, nullptr, nullptr, nullptr, nullptr, nullptr, count_ptr, n_max); }
    }
    { ProfScope ps(e, 4, st); net_stage(e, st, NET_FC, w, nb, nullptr, nullptr, nullptr, nullptr, nullptr, count_ptr, n_max); }
    if (with_heads) {
        ProfScope ps(e, 5, st);
        net_stage(e, st, NET_HEADS, w, nb, nullptr, d.leaf_legal + (size_t)parity * 2 * d.max_games + o, d.leaf_logp + 7 * o, d.leaf_value + o, d.leaf_prior + 7 * o, count_ptr,
                  n_max);
    }
}

// What one lock-step round launches besides the tree step.
struct StepPlan {
    int nbatch;             // leaf batches in use (2 in a two-network arena)
    HeadsPair heads;        // inputs of the heads the tree step computes itself
};

StepPlan make_step_plan(nnb_engine *e, const nnb_search_params &p, const SearchDev &sd) {
    StepPlan plan = {};
    e->policy_heads = p.mode == NNB_MODE_AZ;
    plan.nbatch = sd.wset[0] == sd.wset[1] ? 1 : 2;
    plan.heads.fused = p.evaluator == NNB_EVAL_NET && !(p.flags & NNB_FLAG_SEPARATE_HEADS);
    if (plan.heads.fused)
        for (int b = 0; b < plan.nbatch; ++b) plan.heads.h[b] = heads_input(e, sd.wset[b], b);
    return plan;
}

// One wave busy for `ticks` of the 100 MHz clock; the loop is bounded, every wave leaves it.
__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 20) && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(16);
}

// Do kernels on `a` and `b` run side by side?  HIP spreads a process's streams over a few hardware queues (four by
// default), and two streams that landed on the same queue run their kernels one after the other however independent
// they are.  Which queue a new stream gets depends on the streams the process already has (PyTorch's, a collective
// library's), so it is measured: a 200 us spin on each, the second released by an event of the first.
bool streams_run_side_by_side(hipStream_t a, hipStream_t b) {
    constexpr unsigned long long kTicks = 20000;
    hipEvent_t e0 = nullptr, e1 = nullptr, eb = nullptr;
    bool ok = true;
    float ms = 0.f;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&eb) != hipSuccess) ok = false;
    if (ok) {
        spin_kernel<<<1, 64, 0, a>>>(0);         // the first launch of a kernel loads its code: not part of the measurement
        spin_kernel<<<1, 64, 0, b>>>(0);
        ok = hipStreamSynchronize(b) == hipSuccess;
        ok = ok && hipEventRecord(e0, a) == hipSuccess && hipStreamWaitEvent(b, e0, 0) == hipSuccess;
        if (ok) {
            spin_kernel<<<1, 64, 0, a>>>(kTicks);
            spin_kernel<<<1, 64, 0, b>>>(kTicks);
            ok = hipEventRecord(eb, b) == hipSuccess && hipStreamWaitEvent(a, eb, 0) == hipSuccess &&
                 hipEventRecord(e1, a) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                 hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (eb) (void)hipEventDestroy(eb);
    (void)hipGetLastError();
    return !ok || ms < 0.2f * 1.6f;          // a failed measurement changes nothing
}

// Before the first pair of chains (or of networks) on the caller's stream `st`: if the engine's second stream shares a
// hardware queue with it, other streams are created until one does not (the rejected ones are kept until then, so
// that the next one lands elsewhere).  Nothing of the engine is in flight on the second stream at this point.
#ifndef NNB_STREAM_PROBE
#define NNB_STREAM_PROBE 1               // 0: keep whatever stream was created first (timing comparisons only)
#endif
void ensure_side_stream(nnb_engine *e, hipStream_t st) {
    if (!NNB_STREAM_PROBE) return;
    if (!e->side_stream || (e->side_checked && e->side_checked_for == st)) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
    std::vector<hipStream_t> rejected;
    int tries = 0;
    for (; tries < 8 && !streams_run_side_by_side(st, e->side_stream); ++tries) {
        hipStream_t next = nullptr;
        if (hipStreamCreateWithFlags(&next, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        rejected.push_back(e->side_stream);
        e->side_stream = next;
    }
    for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
    if (getenv("NNB_DEBUG_STREAMS")) fprintf(stderr, "nnb: second stream settled after %d replacement(s)\n", tries);
    e->side_replaced += tries;
    e->side_checked = true;
    e->side_checked_for = st;
}

// One lock-step round: the tree step of every live game, then (unless it is the closing step of a
// move) the evaluation of the leaves it batched.
void lockstep_step(nnb_engine *e, hipStream_t st, const nnb_search_params &p, const SearchDev &sd, const StepPlan &plan,
                   int parity, bool evaluate, const RoundDev *turnover = nullptr) {
    const EngineDev &d = e->dev;
    { ProfScope ps(e, 0, st); launch_tree_step(st, d, sd, parity, plan.heads, turnover); }
    if (!evaluate) return;
    if (plan.nbatch == 2 && p.evaluator == NNB_EVAL_NET && e->side_stream) {
        // two networks: the second batch's forward pass runs on the engine's side stream, next to the
        // first one's, and joins before the next tree step
        e->shared_chip = true;
        (void)hipEventRecord(e->fork_event, st);
        (void)hipStreamWaitEvent(e->side_stream, e->fork_event, 0);
        forward_leaves(e, e->side_stream, sd.wset[1], 1, parity, d.n_games, !plan.heads.fused);
        (void)hipEventRecord(e->join_event, e->side_stream);
        forward_leaves(e, st, sd.wset[0], 0, parity, d.n_games, !plan.heads.fused);
        (void)hipStreamWaitEvent(st, e->join_event, 0);
        e->shared_chip = false;
        return;
    }
    for (int b = 0; b < plan.nbatch; ++b) {
        if (p.evaluator == NNB_EVAL_NET) forward_leaves(e, st, sd.wset[b], b, parity, d.n_games, !plan.heads.fused);
        else launch_fixed_eval(st, d, sd, b, parity);
    }
}

// One move for every live game: S simulations, then the root readout.
int run_move(nnb_engine *e, hipStream_t st, const nnb_search_params &p, int32_t *visits_out,
             int8_t *action_out, double *root_value_out, const RoundDev *round,
             bool noise_given, long long *steps) {
    const int selfplay = round != nullptr;
    SearchDev sd = to_dev(p);
    const EngineDev &d = e->dev;
    e->prof_stride = p.sims / kProfSamples + 1;
    const bool az = p.mode == NNB_MODE_AZ;
    if (!az) { sd.dirichlet_eps = 0.0; sd.dirichlet_alpha = 0.0; }
    int make_noise = 0;
    if (az && sd.dirichlet_eps > 0.0 && !noise_given) {
        if (!(sd.dirichlet_alpha > 0.0)) return fail(NNB_ERR_ARG, "dirichlet_eps > 0 needs dirichlet_alpha > 0 or root_noise");
        make_noise = 1;
    }
    launch_tree_init(st, d, sd, make_noise, selfplay);
    // the fused kernel keeps the node pool of this search in LDS: all of it, or, when that is more than a game's
    // share of the CU, its first nodes (the rest in the game's pool in memory: same results)
    if (p.evaluator == NNB_EVAL_NET || (p.flags & NNB_FLAG_LOCKSTEP)) {
        const StepPlan plan = make_step_plan(e, p, sd);
        if (plan.nbatch == 2 && p.evaluator == NNB_EVAL_NET) ensure_side_stream(e, st);
        for (int s = 0; s <= p.sims; ++s) lockstep_step(e, st, p, sd, plan, s & 1, s < p.sims);
        *steps += p.sims;
    } else {
        const FusedLds plan = fused_lds_plan(p.sims, az, kFusedLdsShare);
        if (int rc = ensure_tree_fused_lds(plan.bytes)) return rc;
        { ProfScope ps(e, 0, st); launch_tree_fused(st, d, sd, plan); }
        *steps += 1;
    }
    { ProfScope ps(e, 6, st); launch_readout(st, d, sd, visits_out, action_out, root_value_out, round); }
    HIP_TRY(hipGetLastError());
    return NNB_OK;
}

int collect_stats(nnb_engine *e, hipStream_t st, nnb_round_stats *out, long long steps, long long rounds, int n,
                  int chains = 1) {
    HIP_TRY(hipMemsetAsync(e->dev.stats64, 0, 8 * sizeof(unsigned long long), st));
    launch_stats_reduce(st, e->dev);
    HIP_TRY(hipMemcpyAsync(e->host_stats, e->dev.stats64, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(e->host_counters + 8, e->dev.counters + kSyncErr, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    prof_drain(e);
    if (e->host_counters[8])
        return fail(NNB_ERR_HIP, "%d block(s) of the linear layer waited in vain for their partner's partial sums: the "
                    "evaluations of this call are invalid", e->host_counters[8]);
    if (out) {
        out->games = n;
        out->depth_sum = (int64_t)e->host_stats[0];
        out->terminal_hits = (int64_t)e->host_stats[1];
        out->net_leaves = (int64_t)e->host_stats[2];
        out->sims = (int64_t)e->host_stats[3];
        out->plies = (int64_t)e->host_stats[4];
        out->cache_hits = (int64_t)e->host_stats[5];
        out->steps = steps;
        out->moves_rounds = rounds;
        out->chains = chains;
    }
    return NNB_OK;
}

// Layer outputs in whatever form the engine's arithmetic keeps them -> float32 (nnb_debug_activations).
// kind 0: float32 copy; 1: halves; 2: ReLU(part0 + part1 + bias) of the split linear layer.
__global__ void debug_act_kernel(int kind, const void *src, const float *src2, const float *bias, size_t total,
                                 float *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (kind == 0) out[i] = static_cast<const float *>(src)[i];
    else if (kind == 1) out[i] = (float)static_cast<const _Float16 *>(src)[i];
    else out[i] = fmaxf((static_cast<const float *>(src)[i] + src2[i]) + bias[i & 1023], 0.f);
}

// ---- search state of nnb_search_static: one fixed array per GPU, no allocation ----------------
constexpr int kStaticRoots = NNB_STATIC_MAX_ROOTS, kStaticSims = NNB_STATIC_MAX_SIMS;
constexpr size_t kStaticPool = 8;                   // root + its children: all the fused kernel hands to the readout
constexpr size_t kStaticBytes = 512 * 1024;

}  // namespace

// Every GPU the library runs on holds its own copy of this array at its own address until the process
// ends.  In host code the name is only a handle for hipGetSymbolAddress; kernels get the address it returns.
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

namespace {

struct LogChunk { double v[256]; };                 // a piece of the log table passed as a kernel argument
__global__ void store_log_chunk_kernel(double *dst, LogChunk c, int n) {
    const int i = threadIdx.x;
    if (i < n) dst[i] = c.v[i];
}

struct StaticState {
    bool ready = false;
    void *ptr = nullptr;                            // device address of `scratch` on this GPU
    EngineDev dev;
};
StaticState g_static[kMaxDevices];
std::mutex g_static_mutex;

// First call on a device: resolve the array's address there, lay the search state out in it, clear it and
// fill the log table (values from the host's libm, handed over as kernel arguments: no copy, no allocation).
// The set-up is enqueued on the caller's stream and WAITED FOR before the state counts as ready, so a later
// call on any other stream finds the table filled; a first call under stream capture is refused (its set-up
// would only be recorded into the graph, not run).
int static_state(hipStream_t st, StaticState **out) {
    int device = -1;
    HIP_TRY(hipGetDevice(&device));
    if (device < 0 || device >= kMaxDevices) return fail(NNB_ERR_ARG, "device ordinal %d out of range", device);
    StaticState &s = g_static[device];
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (!s.ready) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        HIP_TRY(hipStreamIsCapturing(st, &cap));
        if (cap != hipStreamCaptureStatusNone)
            return fail(NNB_ERR_STATE, "the first nnb_search_static call on a device sets its search state up and must not "
                        "run under stream capture: call it once eagerly first");
        void *ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr || reinterpret_cast<uintptr_t>(ptr) % kAlign) return fail(NNB_ERR_HIP, "bad address for the static search state");
        Carver c(ptr);
        carve_state(kStaticRoots, kStaticPool, kStaticSims, 0, c, s.dev);
        if (c.off > kStaticBytes) return fail(NNB_ERR_WORKSPACE, "static search state needs %zu bytes, array has %zu", c.off, kStaticBytes);
        HIP_TRY(hipMemsetAsync(ptr, 0, kStaticBytes, st));
        const int n_log = kStaticSims + 8;
        for (int base = 0; base < n_log; base += 256) {
            LogChunk chunk;
            const int n = n_log - base < 256 ? n_log - base : 256;
            for (int i = 0; i < 256; ++i) chunk.v[i] = std::log((double)(base + i + 1));
            store_log_chunk_kernel<<<1, 256, 0, st>>>(s.dev.log_table + base, chunk, n);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
        s.ptr = ptr;
        s.ready = true;
    }
    *out = &s;
    return NNB_OK;
}

// Root noise is drawn on the device when AZ mode asks for it: *make_noise = 1.
int noise_wanted(const nnb_search_params &p, int *make_noise) {
    *make_noise = 0;
    if (p.mode == NNB_MODE_AZ && p.dirichlet_eps > 0.0) {
        if (!(p.dirichlet_alpha > 0.0)) return fail(NNB_ERR_ARG, "dirichlet_eps > 0 needs dirichlet_alpha > 0");
        *make_noise = 1;
    }
    return NNB_OK;
}

SearchDev lockstep_search_dev(const nnb_search_params &p) {
    SearchDev sd = to_dev(p);
    if (p.mode != NNB_MODE_AZ) { sd.dirichlet_eps = 0.0; sd.dirichlet_alpha = 0.0; }
    return sd;
}

// Self-play without a common move boundary: before every tree step the turnover (a kernel of its own when
// root noise has to be drawn, otherwise part of the step) ends the move of the games whose budget is used up
// and starts their next search; the host only looks at the device counters every kChunk rounds.  Runs
// lock-step rounds step0, step0+1, ... (the round's parity selects the leaf buffers) until no slot holds a
// game, or `finish_target` >= 0 games have finished since the round's set-up, or max_steps rounds have run in
// this call.  *steps_out = rounds run here, *live_out = occupied slots, *finished_out = games finished since
// the set-up.
int run_free_rounds(nnb_engine *e, hipStream_t st, const nnb_search_params &params, const RoundDev &round,
                    int make_noise, long long step0, long long max_steps, long long finish_target,
                    long long *steps_out, int *live_out, long long *finished_out, int *chains_out) {
    const nnb_search_params *p = &params;
    SearchDev sd = lockstep_search_dev(params);
    const int slots = e->dev.n_games;
    StepPlan plan = make_step_plan(e, params, sd);
#ifndef NNB_LOOK_ROUNDS
#define NNB_LOOK_ROUNDS 64
#endif
    constexpr int kChunk = NNB_LOOK_ROUNDS;             // lock-step rounds between two looks of the host at the counters
    e->prof_stride = kChunk / kProfSamples + 1;
    long long steps = 0;
    static_assert(kFinished == kLive + 1 && kSyncErr == kLive + 2, "the host reads the three counters with one copy");
    HIP_TRY(hipMemcpyAsync(e->host_counters, e->dev.counters + kLive, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    int live = e->host_counters[0];
    long long finished = (unsigned)e->host_counters[1];
    // occupied slots never increase between two looks (a slot is refilled at the moment it empties, or not at all)
    e->row_bound = live > 0 ? live : 1;
    auto go_on = [&]() { return live > 0 && steps < max_steps && (finish_target < 0 || finished < finish_target); };
    auto look = [&]() -> hipError_t {
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
        err = hipMemcpyAsync(e->host_counters, e->dev.counters + kLive, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (err != hipSuccess) return err;
        err = hipStreamSynchronize(st);
        if (err != hipSuccess) return err;
        prof_drain(e);
        live = e->host_counters[0];
        finished = (unsigned)e->host_counters[1];
        e->row_bound = live > 0 ? live : 1;
        return hipSuccess;
    };
    // One network and enough slots: the slots are split in two halves, each with a leaf batch, activation
    // buffers and a stream of its own, and each half runs its own tree step -> forward pass chain.  The two
    // chains are not synchronised with each other, so one half's tree step runs under the other half's forward
    // pass, and the CUs that the last blocks of one kernel leave idle take blocks of the other half's kernels.
    // Games do not interact, so the records are the same byte for byte.
    // Measured on 4096 slots x 200 simulations: +5 % games/s with the float32-faithful kernels (one block per CU,
    // long tails), -3 % with the half-input kernels (two blocks per CU already overlap), so those keep one chain.
    const bool wanted = (p->flags & NNB_FLAG_TWO_STREAMS) ||
                        (slots >= kHalvesMinSlots && e->cfg.net_precision != NNB_NET_F16);
    const bool halves = plan.nbatch == 1 && slots >= 2 * kStepGamesHost && wanted && !(p->flags & NNB_FLAG_ONE_STREAM);
    e->shared_chip = halves;
    if (halves || (plan.nbatch == 2 && p->evaluator == NNB_EVAL_NET)) ensure_side_stream(e, st);
    if (halves) {
        const int mid = (slots / 2 + kStepGamesHost - 1) / kStepGamesHost * kStepGamesHost;
        const GameSpan span[2] = {{0, mid, 0}, {mid, slots, 1}};
        if (plan.heads.fused) plan.heads.h[1] = heads_input(e, sd.wset[0], 1);
        hipStream_t hs[2] = {st, e->side_stream};
        while (go_on()) {
            (void)hipEventRecord(e->fork_event, st);
            (void)hipStreamWaitEvent(e->side_stream, e->fork_event, 0);
            for (int k = 0; k < kChunk; ++k, ++steps) {
                const int parity = (int)((step0 + steps) & 1);
                for (int b = 0; b < 2; ++b) {
                    if (make_noise) launch_turnover(hs[b], e->dev, sd, round, make_noise, &span[b]);
                    { ProfScope ps(e, 0, hs[b]); launch_tree_step(hs[b], e->dev, sd, parity, plan.heads, make_noise ? nullptr : &round, &span[b]); }
                    if (p->evaluator == NNB_EVAL_NET)
                        forward_leaves(e, hs[b], sd.wset[0], b, parity, span[b].end - span[b].first, !plan.heads.fused);
                    else
                        launch_fixed_eval(hs[b], e->dev, sd, b, parity);
                }
            }
            (void)hipEventRecord(e->join_event, e->side_stream);
            (void)hipStreamWaitEvent(st, e->join_event, 0);
            HIP_TRY(look());
            if (e->host_counters[2]) break;
        }
    }
    while (!halves && go_on() && !e->host_counters[2]) {
        for (int k = 0; k < kChunk; ++k, ++steps) {
            if (make_noise) launch_turnover(st, e->dev, sd, round, make_noise);
            lockstep_step(e, st, params, sd, plan, (int)((step0 + steps) & 1), true, make_noise ? nullptr : &round);
        }
        HIP_TRY(look());
    }
    e->shared_chip = false;
    e->row_bound = 0;
    if (e->host_counters[2])
        return fail(NNB_ERR_HIP, "%d block(s) of the linear layer waited in vain for their partner's partial sums: the "
                    "evaluations of this call are invalid", e->host_counters[2]);
    *steps_out = steps; *live_out = live; *finished_out = finished; *chains_out = halves ? 2 : 1;
    return NNB_OK;
}

}  // namespace

extern "C" {

const char *nnb_version(void) { return "nnb 0.1 (gfx950)"; }
const char *nnb_last_error(void) { return g_err; }

int nnb_workspace_bytes(const nnb_config *cfg, size_t *bytes_out) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!bytes_out) return fail(NNB_ERR_ARG, "bytes_out is null");
    Carver c(nullptr);
    EngineDev d; NetBuffers b; std::vector<NetWeights> w;
    carve(*cfg, c, d, b, w);
    *bytes_out = c.off;
    return NNB_OK;
}

int nnb_engine_create(const nnb_config *cfg, void *workspace, size_t workspace_bytes, void *stream,
                      nnb_engine **engine_out) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!workspace || !engine_out) return fail(NNB_ERR_ARG, "workspace / engine_out is null");
    if (reinterpret_cast<uintptr_t>(workspace) % kAlign)
        return fail(NNB_ERR_WORKSPACE, "workspace must be %zu-byte aligned", kAlign);
    size_t need = 0;
    nnb_workspace_bytes(cfg, &need);
    if (workspace_bytes < need)
        return fail(NNB_ERR_WORKSPACE, "workspace has %zu bytes, %zu needed", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(cfg->device));
    struct Guard {                          // frees a half-built engine when a step below fails
        nnb_engine *e;
        ~Guard() { if (e) nnb_engine_destroy(e); }
    } guard{new (std::nothrow) nnb_engine()};
    nnb_engine *e = guard.e;
    if (!e) return fail(NNB_ERR_ARG, "out of host memory");
    e->cfg = *cfg;
    Carver c(workspace);
    const size_t state_bytes = carve(*cfg, c, e->dev, e->buf, e->weights);
    e->buf.shared_chip = 0;
    e->weights_loaded.assign(cfg->n_weight_sets, 0);
    HIP_TRY(hipMemsetAsync(workspace, 0, state_bytes, st));
    // log(n + 1) from the host's libm: the same function CPython's math.log calls (mcts.py:29)
    std::vector<double> logt(cfg->max_sims + 8);
    for (size_t n = 0; n < logt.size(); ++n) logt[n] = std::log((double)(n + 1));
    HIP_TRY(hipMemcpyAsync(e->dev.log_table, logt.data(), logt.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->host_counters), 64 * sizeof(int32_t), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->host_stats), 32 * sizeof(unsigned long long), hipHostMallocDefault));
    {
        HIP_TRY(hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e->fork_event, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->join_event, hipEventDisableTiming));
    }
    HIP_TRY(prepare_net_kernels());
    HIP_TRY(prepare_tree_step());
    HIP_TRY(prepare_net_kernels_f16());
    HIP_TRY(prepare_net_kernels_split());
    HIP_TRY(hipStreamSynchronize(st));
    guard.e = nullptr;
    *engine_out = e;
    return NNB_OK;
}

int nnb_engine_destroy(nnb_engine *e) {
    if (!e) return NNB_OK;
    if (e->prof_events)
        for (auto &p : e->prof)
            for (int i = 0; i < kProfSamples; ++i) { (void)hipEventDestroy(p.start[i]); (void)hipEventDestroy(p.stop[i]); }
    if (e->side_stream) (void)hipStreamDestroy(e->side_stream);
    if (e->fork_event) (void)hipEventDestroy(e->fork_event);
    if (e->join_event) (void)hipEventDestroy(e->join_event);
    if (e->host_counters) (void)hipHostFree(e->host_counters);
    if (e->host_stats) (void)hipHostFree(e->host_stats);
    delete e;
    return NNB_OK;
}

int nnb_load_weights(nnb_engine *e, int weight_set, void *stream, const float *conv0_w, const float *conv0_b,
                     const float *conv2_w, const float *conv2_b, const float *conv4_w, const float *conv4_b,
                     const float *fc_w, const float *fc_b, const float *policy_w, const float *policy_b,
                     const float *value_w, const float *value_b) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    if (weight_set < 0 || weight_set >= e->cfg.n_weight_sets)
        return fail(NNB_ERR_ARG, "weight_set %d out of range [0, %d)", weight_set, e->cfg.n_weight_sets);
    if (!conv0_w || !conv0_b || !conv2_w || !conv2_b || !conv4_w || !conv4_b || !fc_w || !fc_b || !policy_w ||
        !policy_b || !value_w || !value_b)
        return fail(NNB_ERR_ARG, "a weight pointer is null");
    hipStream_t st = static_cast<hipStream_t>(stream);
    launch_repack(st, e->weights[weight_set], conv0_w, conv0_b, conv2_w, conv2_b, conv4_w, conv4_b, fc_w, fc_b,
                  policy_w, policy_b, value_w, value_b);
    if (e->cfg.net_precision == NNB_NET_F16) launch_repack_f16(st, e->weights[weight_set], conv2_w, conv4_w, fc_w);
    if (e->cfg.net_precision == NNB_NET_F32S)
        launch_repack_split(st, e->weights[weight_set], conv0_w, conv0_b, conv2_w, conv4_w, fc_w);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    e->weights_loaded[weight_set] = 1;
    return NNB_OK;
}

int nnb_game_ops(void *stream, int32_t n, const uint64_t *in, const int8_t *players, const int8_t *actions,
                 uint64_t *out, uint8_t *legal_out, uint8_t *status_out) {
    if (n < 0) return fail(NNB_ERR_ARG, "n < 0");
    if (n == 0) return NNB_OK;
    if (!in) return fail(NNB_ERR_ARG, "colour_boards_in is null");
    if (actions && !players) return fail(NNB_ERR_ARG, "players is null but actions is given");
    launch_game_ops(static_cast<hipStream_t>(stream), n, reinterpret_cast<const u64 *>(in), players, actions,
                    reinterpret_cast<u64 *>(out), legal_out, status_out);
    HIP_TRY(hipGetLastError());
    return NNB_OK;
}

int nnb_minimax(void *stream, int32_t n, const uint64_t *colour_boards, const int8_t *movers, int32_t depth,
                int8_t *column_out, int32_t *score_out) {
    if (n < 0) return fail(NNB_ERR_ARG, "n < 0");
    if (n == 0) return NNB_OK;
    if (!colour_boards || !movers) return fail(NNB_ERR_ARG, "null pointer");
    if (depth < 0 || depth > 8) return fail(NNB_ERR_ARG, "depth %d out of range [0, 8]", depth);
    launch_minimax(static_cast<hipStream_t>(stream), n, reinterpret_cast<const u64 *>(colour_boards), movers, depth,
                   column_out, score_out);
    HIP_TRY(hipGetLastError());
    return NNB_OK;
}

int nnb_encode_planes(void *stream, int32_t n, const uint64_t *side_boards, float *planes_out) {
    if (n < 0) return fail(NNB_ERR_ARG, "n < 0");
    if (n == 0) return NNB_OK;
    if (!side_boards || !planes_out) return fail(NNB_ERR_ARG, "null pointer");
    launch_encode_planes(static_cast<hipStream_t>(stream), n, reinterpret_cast<const u64 *>(side_boards), planes_out);
    HIP_TRY(hipGetLastError());
    return NNB_OK;
}

int nnb_ucb_scores(nnb_engine *e, void *stream, int32_t n, const double *value_sum, const int32_t *visits,
                   const int32_t *parent_visits, double c_puct, double *scores_out) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    if (n < 0) return fail(NNB_ERR_ARG, "n < 0");
    if (n == 0) return NNB_OK;
    if (!value_sum || !visits || !parent_visits || !scores_out) return fail(NNB_ERR_ARG, "null pointer");
    launch_ucb_scores(static_cast<hipStream_t>(stream), n, value_sum, visits, parent_visits, c_puct,
                      e->dev.log_table, e->cfg.max_sims + 8, scores_out);
    HIP_TRY(hipGetLastError());
    return NNB_OK;
}

int nnb_net_forward(nnb_engine *e, int weight_set, void *stream, int32_t n, const uint64_t *side_boards,
                    const uint8_t *legal_masks, float *log_policy_out, float *value_out, float *priors_out) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    if (n < 0 || n > e->cfg.max_games) return fail(NNB_ERR_ARG, "n = %d out of range [0, %d]", n, e->cfg.max_games);
    if (weight_set < 0 || weight_set >= e->cfg.n_weight_sets) return fail(NNB_ERR_ARG, "weight_set out of range");
    if (!e->weights_loaded[weight_set]) return fail(NNB_ERR_STATE, "weight set %d has not been loaded", weight_set);
    if (n == 0) return NNB_OK;
    if (!side_boards) return fail(NNB_ERR_ARG, "side_boards is null");
    if (priors_out && !legal_masks) return fail(NNB_ERR_ARG, "priors_out needs legal_masks");
    e->stream_on = false;                                // the activations of a seated stream's pending leaves are overwritten
    hipStream_t st = static_cast<hipStream_t>(stream);
    const NetWeights &w = e->weights[weight_set];
    const u64 *side = reinterpret_cast<const u64 *>(side_boards);
    NetBuffers nb = batch_buffers(e, 0);
    nb.heads_all = 1;
    nb.write_hidden = 1;                                 // for nnb_debug_activations
    { ProfScope ps(e, 1, st); net_stage(e, st, NET_CONV0, w, nb, side, nullptr, nullptr, nullptr, nullptr, nullptr, n); }
    if (e->cfg.net_precision == NNB_NET_F32S) {         // one launch; the middle activations are copied out for nnb_debug_activations
        ProfScope ps(e, 2, st); net_stage(e, st, NET_CONV_PAIR_MID, w, nb, side, nullptr, nullptr, nullptr, nullptr, nullptr, n);
    } else {
        { ProfScope ps(e, 2, st); net_stage(e, st, NET_CONV2, w, nb, side, nullptr, nullptr, nullptr, nullptr, nullptr, n); }
        { ProfScope ps(e, 3, st); net_stage(e, st, NET_CONV4, w, nb, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, n); }
    }
    { ProfScope ps(e, 4, st); net_stage(e, st, NET_FC, w, nb, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, n); }
    { ProfScope ps(e, 5, st); net_stage(e, st, NET_HEADS, w, nb, nullptr, legal_masks, log_policy_out, value_out, priors_out, nullptr, n); }
    HIP_TRY(hipGetLastError());
    return NNB_OK;
}

int nnb_net_forward_planes(nnb_engine *e, int weight_set, void *stream, int32_t n, const float *planes,
                           float *log_policy_out, float *value_out) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    if (n < 0 || n > e->cfg.max_games) return fail(NNB_ERR_ARG, "n = %d out of range [0, %d]", n, e->cfg.max_games);
    if (n == 0) return NNB_OK;
    if (!planes) return fail(NNB_ERR_ARG, "planes is null");
    hipStream_t st = static_cast<hipStream_t>(stream);
    launch_planes_to_bits(st, n, planes, e->dev.leaf_side);
    return nnb_net_forward(e, weight_set, stream, n, reinterpret_cast<const uint64_t *>(e->dev.leaf_side), nullptr,
                           log_policy_out, value_out, nullptr);
}

int nnb_debug_activations(nnb_engine *e, int weight_set, void *stream, int layer, int32_t n, float *out) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    if (n < 0 || n > e->cfg.max_games) return fail(NNB_ERR_ARG, "n = %d out of range [0, %d]", n, e->cfg.max_games);
    if (weight_set < 0 || weight_set >= e->cfg.n_weight_sets) return fail(NNB_ERR_ARG, "weight_set out of range");
    if (layer < 2 || layer > 4) return fail(NNB_ERR_ARG, "layer %d: 2 and 3 are the 128->128 convolutions, 4 the linear layer", layer);
    if (n == 0) return NNB_OK;
    if (!out) return fail(NNB_ERR_ARG, "out is null");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const NetBuffers &b = e->buf;
    const int prec = e->cfg.net_precision;
    if (layer == 4) {
        const size_t total = (size_t)n * 1024;
        const unsigned blocks = (unsigned)((total + 255) / 256);
        if (prec == NNB_NET_F32 || prec == NNB_NET_F32S) debug_act_kernel<<<blocks, 256, 0, st>>>(0, b.fc, nullptr, nullptr, total, out);
        else debug_act_kernel<<<blocks, 256, 0, st>>>(2, b.fcpart, b.fcpart + (size_t)b.max_rows * 1024,
                                                      e->weights[weight_set].fc_b, total, out);
    } else {
        const int cells = layer == 2 ? 72 : 90;
        const float *act = layer == 2 ? b.act2 : b.act3;
        const size_t total = (size_t)n * cells * 128;
        const unsigned blocks = (unsigned)((total + 255) / 256);
        if (prec == NNB_NET_F32S) launch_split_to_f32(st, act, layer == 2 ? b.act2_inv : b.act3_inv, cells, n, out);
        else debug_act_kernel<<<blocks, 256, 0, st>>>(prec == NNB_NET_F32 ? 0 : 1, act, nullptr, nullptr, total, out);
    }
    HIP_TRY(hipGetLastError());
    return NNB_OK;
}

int nnb_search(nnb_engine *e, void *stream, const nnb_search_params *params, int32_t n,
               const uint64_t *root_colour_boards, const int8_t *root_players, const int8_t *side_players,
               const float *root_noise, int32_t *visits_out, int8_t *action_out, double *root_value_out, nnb_round_stats *stats_out) {
    if (e && params && n == 0) {                         // no roots: a no-op like the other entry points
        if (stats_out) memset(stats_out, 0, sizeof *stats_out);
        return NNB_OK;
    }
    int rc = check_params(e, params, n);
    if (rc) return rc;
    if (!root_colour_boards || !root_players) return fail(NNB_ERR_ARG, "root boards / players is null");
    hipStream_t st = static_cast<hipStream_t>(stream);
    e->dev.n_games = n;
    e->stream_on = false;
    launch_search_init(st, e->dev, reinterpret_cast<const u64 *>(root_colour_boards), root_players, side_players,
                       params->rng_game_base, root_noise);
    long long steps = 0;
    rc = run_move(e, st, *params, visits_out, action_out, root_value_out, nullptr, root_noise != nullptr, &steps);
    if (rc) return rc;
    return collect_stats(e, st, stats_out, steps, 1, n);
}

int nnb_selfplay_round(nnb_engine *e, void *stream, const nnb_search_params *params, int32_t n_games,
                       int32_t first_game_id, const uint64_t *start_colour_boards, const int8_t *start_players,
                       nnb_trajectory *trajectories_out, nnb_round_stats *stats_out) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    if (n_games < 0) return fail(NNB_ERR_ARG, "n_games < 0");
    if (n_games == 0) {                                  // nothing to play: a no-op like the other entry points
        if (stats_out) memset(stats_out, 0, sizeof *stats_out);
        return NNB_OK;
    }
    const int slots = n_games < e->cfg.max_games ? n_games : e->cfg.max_games;
    int rc = check_params(e, params, slots);
    if (rc) return rc;
    if (!trajectories_out) return fail(NNB_ERR_ARG, "trajectories_out is null");
    if ((start_colour_boards == nullptr) != (start_players == nullptr))
        return fail(NNB_ERR_ARG, "start boards and start players must be given together");
    hipStream_t st = static_cast<hipStream_t>(stream);
    e->dev.n_games = slots;
    const RoundDev round = {reinterpret_cast<const u64 *>(start_colour_boards), start_players, trajectories_out, n_games,
                            first_game_id, n_games, n_games, nullptr};
    e->stream_on = false;
    HIP_TRY(hipMemsetAsync(e->dev.counters, 0, 32 * sizeof(int32_t), st));      // live count and refill count
    launch_selfplay_init(st, e->dev, round);
    HIP_TRY(hipMemcpyAsync(e->host_counters, e->dev.counters, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    long long steps = 0, rounds = 0;
    int live = e->host_counters[0];
    const bool lockstep = params->evaluator == NNB_EVAL_NET || (params->flags & NNB_FLAG_LOCKSTEP);
    if (lockstep && !(params->flags & NNB_FLAG_SYNC_MOVES)) {
        int make_noise = 0;
        rc = noise_wanted(*params, &make_noise);
        if (rc) return rc;
        SearchDev sd = lockstep_search_dev(*params);
        launch_tree_init(st, e->dev, sd, make_noise, 1);
        long long max_steps = ((long long)n_games / (slots > 0 ? slots : 1) + 2) * (kMaxPlies + 1) * (params->sims + 1);
        if (params->round_limit > 0 && params->round_limit < max_steps) max_steps = params->round_limit;
        int chains = 1;
        long long finished = 0;
        rc = run_free_rounds(e, st, *params, round, make_noise, 0, max_steps, -1, &steps, &live, &finished, &chains);
        if (rc) return rc;
        rounds = (steps + params->sims) / (params->sims + 1);
        rc = collect_stats(e, st, stats_out, steps, rounds, n_games, chains);
        if (rc) return rc;
        if (live > 0)
            return fail(NNB_ERR_STATE, "self-play stopped after %lld lock-step rounds with %d games unfinished "
                        "(their records have result = 2)", steps, live);
        return NNB_OK;
    }
    const long long max_rounds = ((long long)n_games / (slots > 0 ? slots : 1) + 2) * (kMaxPlies + 1);
    while (live > 0 && rounds < max_rounds && !(params->round_limit > 0 && steps >= params->round_limit)) {
        HIP_TRY(hipMemsetAsync(e->dev.counters, 0, 16 * sizeof(int32_t), st));
        rc = run_move(e, st, *params, nullptr, nullptr, nullptr, &round, false, &steps);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(e->host_counters, e->dev.counters, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        prof_drain(e);
        live = e->host_counters[0];
        ++rounds;
    }
    rc = collect_stats(e, st, stats_out, steps, rounds, n_games);
    if (rc) return rc;
    if (live > 0)
        return fail(NNB_ERR_STATE, "self-play stopped after %lld move rounds with %d games unfinished "
                    "(their records have result = 2)", rounds, live);
    return NNB_OK;
}

int nnb_selfplay_stream(nnb_engine *e, void *stream, const nnb_search_params *params, int32_t restart,
                        int32_t first_game_id, int32_t min_finished, const uint64_t *start_colour_boards,
                        const int8_t *start_players, int32_t n_starts, nnb_trajectory *record_ring,
                        int32_t *finished_ring, int32_t ring_len, nnb_round_stats *stats_out) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    if (min_finished < 0) return fail(NNB_ERR_ARG, "min_finished < 0");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (restart) {
        const int slots = e->cfg.max_games;
        int rc = check_params(e, params, slots);
        if (rc) return rc;
        if (params->evaluator != NNB_EVAL_NET && !(params->flags & NNB_FLAG_LOCKSTEP))
            return fail(NNB_ERR_ARG, "a stream of games runs lock-step rounds: a fixed evaluator needs NNB_FLAG_LOCKSTEP");
        if (params->flags & NNB_FLAG_SYNC_MOVES) return fail(NNB_ERR_ARG, "a stream of games has no common move boundary (NNB_FLAG_SYNC_MOVES)");
        if (!record_ring) return fail(NNB_ERR_ARG, "record_ring is null");
        if (ring_len < 4 * slots) return fail(NNB_ERR_ARG, "ring_len = %d: the record ring needs at least 4 x max_games = %d records", ring_len, 4 * slots);
        if ((start_colour_boards == nullptr) != (start_players == nullptr))
            return fail(NNB_ERR_ARG, "start boards and start players must be given together");
        if (start_colour_boards && n_starts < 1) return fail(NNB_ERR_ARG, "n_starts < 1");
        e->stream_on = false;
        e->dev.n_games = slots;
        e->stream_round = RoundDev{reinterpret_cast<const u64 *>(start_colour_boards), start_players, record_ring,
                                   0x7fffffff - slots, first_game_id, ring_len, start_colour_boards ? n_starts : 1,
                                   finished_ring};
        int make_noise = 0;
        rc = noise_wanted(*params, &make_noise);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(e->dev.counters, 0, 32 * sizeof(int32_t), st));
        launch_selfplay_init(st, e->dev, e->stream_round);
        launch_tree_init(st, e->dev, lockstep_search_dev(*params), make_noise, 1);
        HIP_TRY(hipGetLastError());
        e->stream_params = *params;
        e->stream_slots = slots;
        e->stream_steps = 0;
        e->stream_finished = 0;
        for (auto &v : e->stream_prev) v = 0;
        e->stream_on = true;
    } else {
        if (!e->stream_on)
            return fail(NNB_ERR_STATE, "no stream of games is seated on this engine (never started, or another call "
                        "has used the engine since): call with restart = 1");
        if (params && memcmp(params, &e->stream_params, sizeof *params) != 0)
            return fail(NNB_ERR_ARG, "the search parameters differ from those the stream was started with");
    }
    if (min_finished > e->stream_round.ring_len - 2 * e->stream_slots)
        return fail(NNB_ERR_ARG, "min_finished = %d: a call's records must fit the ring beside the games in flight "
                    "(at most ring_len - 2 x max_games = %d per call)", min_finished,
                    e->stream_round.ring_len - 2 * e->stream_slots);
    const nnb_search_params &p = e->stream_params;
    e->dev.n_games = e->stream_slots;
    int make_noise = 0;
    (void)noise_wanted(p, &make_noise);
    long long max_steps = ((long long)min_finished / e->stream_slots + 2) * (kMaxPlies + 1) * (p.sims + 1);
    if (p.round_limit > 0 && p.round_limit < max_steps) max_steps = p.round_limit;
    long long steps = 0, finished = e->stream_finished;
    int live = 0, chains = 1;
    int rc = run_free_rounds(e, st, p, e->stream_round, make_noise, e->stream_steps, max_steps,
                             e->stream_finished + min_finished, &steps, &live, &finished, &chains);
    if (rc) { e->stream_on = false; return rc; }
    if (finished - e->stream_finished > e->stream_round.ring_len - e->stream_slots) {
        // more games finished inside this call than the ring holds beside the games in flight (very short searches,
        // or start positions that are mostly finished): records it was about to report have been overwritten
        e->stream_on = false;
        return fail(NNB_ERR_STATE, "%lld games finished in one call but the record ring holds %d: records were overwritten; "
                    "use a longer ring or a smaller min_finished", finished - e->stream_finished, e->stream_round.ring_len);
    }
    e->stream_steps += steps;
    nnb_round_stats total;
    rc = collect_stats(e, st, &total, steps, (steps + p.sims) / (p.sims + 1), 0, chains);
    if (rc) { e->stream_on = false; return rc; }
    long long *prev = e->stream_prev;
    const long long now[6] = {total.depth_sum, total.terminal_hits, total.net_leaves, total.sims, total.plies, total.cache_hits};
    if (stats_out) {
        *stats_out = total;
        stats_out->games = finished - e->stream_finished;
        stats_out->depth_sum = now[0] - prev[0];
        stats_out->terminal_hits = now[1] - prev[1];
        stats_out->net_leaves = now[2] - prev[2];
        stats_out->sims = now[3] - prev[3];
        stats_out->plies = now[4] - prev[4];
        stats_out->cache_hits = now[5] - prev[5];
        stats_out->finished_total = finished;
    }
    for (int k = 0; k < 6; ++k) prev[k] = now[k];
    const long long got = finished - e->stream_finished;
    e->stream_finished = finished;
    if (got < min_finished)
        return fail(NNB_ERR_STATE, "the stream stopped after %lld lock-step rounds with %lld of the %d games asked for finished",
                    steps, got, min_finished);
    return NNB_OK;
}

int nnb_trajectory_samples(void *stream, int32_t n, const nnb_trajectory *records, const int32_t *ply_offsets,
                            float *planes_out, float *probs_out, float *z_out) {
    if (n < 0) return fail(NNB_ERR_ARG, "n < 0");
    if (n == 0) return NNB_OK;
    if (!records || !ply_offsets || !planes_out || !probs_out || !z_out) return fail(NNB_ERR_ARG, "null pointer");
    launch_trajectory_samples(static_cast<hipStream_t>(stream), n, records, ply_offsets, planes_out, probs_out, z_out);
    HIP_TRY(hipGetLastError());
    return NNB_OK;
}

int nnb_dirichlet_noise(void *stream, int32_t n, uint64_t seed, const int32_t *game_ids, const int32_t *move_indices,
                        const uint8_t *legal_masks, double alpha, float *noise_out) {
    if (n < 0) return fail(NNB_ERR_ARG, "n < 0");
    if (n == 0) return NNB_OK;
    if (!game_ids || !move_indices || !legal_masks || !noise_out) return fail(NNB_ERR_ARG, "null pointer");
    if (!(alpha > 0.0) || !std::isfinite(alpha)) return fail(NNB_ERR_ARG, "alpha must be finite and > 0");
    launch_dirichlet_noise(static_cast<hipStream_t>(stream), n, seed, game_ids, move_indices, legal_masks, alpha,
                           noise_out);
    HIP_TRY(hipGetLastError());
    return NNB_OK;
}

int nnb_search_static(void *stream, const nnb_search_params *p, int32_t n, const uint64_t *root_colour_boards,
                      const int8_t *root_players, const int8_t *side_players, const float *root_noise,
                      int32_t *visits_out, int8_t *action_out, double *root_value_out) {
    if (!p) return fail(NNB_ERR_ARG, "params is null");
    if (n < 0 || n > kStaticRoots) return fail(NNB_ERR_ARG, "n = %d out of range [0, NNB_STATIC_MAX_ROOTS = %d]", n, kStaticRoots);
    if (p->sims < 1 || p->sims > kStaticSims)
        return fail(NNB_ERR_ARG, "sims = %d out of range [1, NNB_STATIC_MAX_SIMS = %d]", p->sims, kStaticSims);
    if (p->mode != NNB_MODE_COMPAT && p->mode != NNB_MODE_AZ) return fail(NNB_ERR_ARG, "unknown mode %d", p->mode);
    if (p->evaluator <= NNB_EVAL_NET || p->evaluator > NNB_EVAL_ROLLOUT)
        return fail(NNB_ERR_ARG, "nnb_search_static runs the fixed evaluators only (CONST, HASH_Q8, HASH_F32, ROLLOUT), not %d",
                    p->evaluator);
    if (p->flags != 0) return fail(NNB_ERR_ARG, "nnb_search_static takes no flags (the fused tree kernel only)");
    if (!(p->c_puct >= 0.0) || !std::isfinite(p->c_puct)) return fail(NNB_ERR_ARG, "c_puct must be finite and >= 0");
    if (p->weight_set < 0 || p->weight_set >= 8 || p->weight_set_colour2_plus1 < 0 || p->weight_set_colour2_plus1 > 8)
        return fail(NNB_ERR_ARG, "weight set out of range [0, 8)");
    for (int k = 0; k < 2; ++k)
        if (p->side_of_colour[k] < 0 || p->side_of_colour[k] > 2)
            return fail(NNB_ERR_ARG, "side_of_colour[%d] = %d is not 0 (the mover), 1 or 2", k, p->side_of_colour[k]);
    const bool az = p->mode == NNB_MODE_AZ;
    if (az) {
        if (!(p->dirichlet_eps >= 0.0 && p->dirichlet_eps <= 1.0)) return fail(NNB_ERR_ARG, "dirichlet_eps must be in [0, 1]");
        if (!(p->dirichlet_alpha >= 0.0) || !std::isfinite(p->dirichlet_alpha))
            return fail(NNB_ERR_ARG, "dirichlet_alpha must be finite and >= 0");
    }
    if (n == 0) return NNB_OK;
    if (!root_colour_boards || !root_players) return fail(NNB_ERR_ARG, "root boards / players is null");
    hipStream_t st = static_cast<hipStream_t>(stream);
    StaticState *s = nullptr;
    int rc = static_state(st, &s);
    if (rc) return rc;
    SearchDev sd = to_dev(*p);
    if (!az) { sd.dirichlet_eps = 0.0; sd.dirichlet_alpha = 0.0; }
    int make_noise = 0;
    if (az && sd.dirichlet_eps > 0.0 && !root_noise) {
        if (!(sd.dirichlet_alpha > 0.0)) return fail(NNB_ERR_ARG, "dirichlet_eps > 0 needs dirichlet_alpha > 0 or root_noise");
        make_noise = 1;
    }
    // no pools in memory behind this entry point: the whole pool in LDS
    const FusedLds plan = fused_lds_plan(p->sims, az, kFusedLdsLimit);
    if (plan.split) return fail(NNB_ERR_ARG, "sims = %d needs more than the %zu bytes of LDS of a CU", p->sims, kFusedLdsLimit);
    rc = ensure_tree_fused_lds(plan.bytes);
    if (rc) return rc;
    EngineDev d = s->dev;                   // every pointer in it derives from the address hipGetSymbolAddress returned
    d.n_games = n;
    launch_search_init(st, d, reinterpret_cast<const u64 *>(root_colour_boards), root_players, side_players,
                       p->rng_game_base, root_noise);
    launch_tree_init(st, d, sd, make_noise, 0);
    launch_tree_fused(st, d, sd, plan);
    launch_readout(st, d, sd, visits_out, action_out, root_value_out, nullptr);
    HIP_TRY(hipGetLastError());
    return NNB_OK;
}

int nnb_profile_enable(nnb_engine *e, int on) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    if (on && !e->prof_events) {
        for (auto &p : e->prof)
            for (int i = 0; i < kProfSamples; ++i) {
                HIP_TRY(hipEventCreate(&p.start[i]));
                HIP_TRY(hipEventCreate(&p.stop[i]));
            }
        e->prof_events = true;
    }
    e->prof_on = on != 0;
    return NNB_OK;
}

int nnb_profile_clocks(nnb_engine *e, void *stream, double *mhz_out) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    if (!mhz_out) return fail(NNB_ERR_ARG, "mhz_out is null");
    unsigned long long h[8];
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(h, e->buf.clk, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < 2; ++k) {
        const unsigned long long *c = h + 4 * k;
        // the wall clock ticks at 100 MHz
        mhz_out[k] = (c[3] > c[1] && c[2] > c[0]) ? (double)(c[2] - c[0]) / (double)(c[3] - c[1]) * 100.0 : 0.0;
    }
    return NNB_OK;
}

int nnb_debug_second_stream(nnb_engine *e, void *stream, int *replaced_out) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    hipStream_t st = static_cast<hipStream_t>(stream);
    e->side_checked = false;
    ensure_side_stream(e, st);
    if (replaced_out) *replaced_out = e->side_replaced;
    const bool beside = streams_run_side_by_side(st, e->side_stream);
    HIP_TRY(hipGetLastError());
    return beside ? 1 : 0;
}

int nnb_profile_reset(nnb_engine *e) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    for (auto &p : e->prof) { p.used = 0; p.total_ms = 0.0; p.launches = 0; p.seen = 0; }
    return NNB_OK;
}

int nnb_profile_get(nnb_engine *e, int which, double *avg_ms_out, int64_t *launches_out) {
    if (!e) return fail(NNB_ERR_ARG, "engine is null");
    if (which < 0 || which >= kProfClasses) return fail(NNB_ERR_ARG, "class %d out of range", which);
    const ProfClass &p = e->prof[which];
    if (avg_ms_out) *avg_ms_out = p.launches ? p.total_ms / (double)p.launches : 0.0;
    if (launches_out) *launches_out = p.launches;
    return NNB_OK;
}

}  // extern "C"
