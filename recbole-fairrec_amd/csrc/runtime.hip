// Error string plumbing, version, and the optional per-kernel HIP-event profiler of libfairrec_hip.so.
#include <stdarg.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include <unordered_map>
#include <vector>
#include <mutex>

#include "common.hpp"
#include "kernels.hpp"

namespace fr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- profiler: hipEvent pairs recorded on the launch stream around each kernel --------------------
static const char* const kKernelNames[K_COUNT] = {
    "sort_segments_kernel", "focf_gather_kernel",     "focf_fair_kernel",    "focf_nonparity_kernel",
    "focf_finalize_kernel", "focf_backward_adam_kernel", "table_flush_kernel", "table_gather_kernel",
    "adam_dense_kernel",    "table_gather_train_kernel", "table_apply_grad_kernel",
    "bucket_by_owner_kernel", "unbucket_rows_kernel",     "bucket_rows_kernel",
    "focf_shard_score_kernel", "focf_shard_grads_kernel", "linear_fwd_kernel",
    "linear_bwd_input_kernel", "linear_bwd_weight_kernel", "nfcf_bce_kernel",
    "bn_fwd_kernel", "bn_bwd_kernel", "rowdot_kernel", "bpr_kernel", "spmm_csr_kernel", "row_gather_scatter_kernel",
    "sample_negatives_kernel", "focf_step_kernel", "focf_lpt_kernel", "focf_stage_kernel"};

struct ProfState {
    bool on = false;
    std::mutex mu;
    std::vector<hipEvent_t> pool;
    struct Pair {
        hipEvent_t a, b;
        int kind;
    };
    std::vector<Pair> open;
    double total_ms[K_COUNT] = {0};
    long long count[K_COUNT] = {0};
    double work[K_COUNT] = {0};       // algorithmic work of the launches of a kind: FLOP for the dense layers, bytes for SpMM
};
static ProfState g_prof;

bool prof_on() { return g_prof.on; }

void prof_work(int kind, double amount) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.work[kind] += amount;
}

static hipEvent_t take_event() {
    if (!g_prof.pool.empty()) {
        hipEvent_t e = g_prof.pool.back();
        g_prof.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

bool prof_take(int kind, hipEvent_t* start, hipEvent_t* stop) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (!g_prof.on) return false;
    ProfState::Pair p{take_event(), take_event(), kind};
    if (!p.a || !p.b) return false;
    g_prof.open.push_back(p);
    *start = p.a;
    *stop = p.b;
    return true;
}

static void prof_drain() {
    for (auto& p : g_prof.open) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_prof.total_ms[p.kind] += ms;
            g_prof.count[p.kind] += 1;
        }
        g_prof.pool.push_back(p.a);
        g_prof.pool.push_back(p.b);
    }
    g_prof.open.clear();
}

SideStream* side_stream() {
    static SideStream ss;
    static int state = 0;  // 0 = not tried, 1 = ready, -1 = disabled / failed
    if (state == 0) {
        const char* off = getenv("FAIRREC_NO_OVERLAP");
        state = -1;
        if (!(off && off[0] == '1') &&
            hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess)
            state = 1;
    }
    return state == 1 ? &ss : nullptr;
}

namespace {
std::mutex g_side_mu;
std::unordered_map<const void*, hipEvent_t> g_side_pending;
std::vector<hipEvent_t> g_side_pool;
}  // namespace

int side_mark(const void* ws) {
    SideStream* ss = side_stream();
    if (!ss) return FR_OK;
    std::lock_guard<std::mutex> lk(g_side_mu);
    hipEvent_t ev;
    auto it = g_side_pending.find(ws);
    if (it != g_side_pending.end()) {
        ev = it->second;   // a sort of the same workspace that nobody consumed: later work on the side stream follows it
    } else if (!g_side_pool.empty()) {
        ev = g_side_pool.back();
        g_side_pool.pop_back();
    } else {
        FR_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    g_side_pending[ws] = ev;
    FR_CHECK_HIP(hipEventRecord(ev, ss->stream));
    return FR_OK;
}

int side_join(const void* ws, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    auto it = g_side_pending.find(ws);
    if (it == g_side_pending.end()) return FR_OK;
    hipEvent_t ev = it->second;
    g_side_pending.erase(it);
    g_side_pool.push_back(ev);
    FR_CHECK_HIP(hipStreamWaitEvent(stream, ev, 0));
    return FR_OK;
}
}  // namespace fr

using namespace fr;

// the stream the library's own look-ahead work runs on (NULL: none) -- for callers whose allocator must know every stream
// a buffer is used on before it hands the buffer's memory to somebody else
extern "C" void* fr_side_stream_handle(void) {
    SideStream* ss = side_stream();
    return ss ? (void*)ss->stream : nullptr;
}

extern "C" int fr_version(void) { return 1; }
extern "C" const char* fr_last_error(void) { return g_err; }

extern "C" int fr_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (!on) prof_drain();
    g_prof.on = on != 0;
    return FR_OK;
}

extern "C" int fr_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    prof_drain();
    for (int k = 0; k < K_COUNT; ++k) {
        g_prof.total_ms[k] = 0;
        g_prof.count[k] = 0;
        g_prof.work[k] = 0;
    }
    return FR_OK;
}

extern "C" int fr_prof_read_work(int kind, double* work) {
    FR_CHECK_ARG(kind >= 0 && kind < K_COUNT && work, "fr_prof_read_work: bad argument");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    *work = g_prof.work[kind];
    return FR_OK;
}

extern "C" int fr_prof_kernel_count(void) { return K_COUNT; }

extern "C" const char* fr_prof_kernel_name(int kind) { return kind >= 0 && kind < K_COUNT ? kKernelNames[kind] : ""; }

extern "C" int fr_prof_read(int kind, double* total_ms, int64_t* count) {
    FR_CHECK_ARG(kind >= 0 && kind < K_COUNT && total_ms && count, "fr_prof_read: bad argument");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    prof_drain();
    *total_ms = g_prof.total_ms[kind];
    *count = g_prof.count[kind];
    return FR_OK;
}

// ---- running loss total of a step loop, with a sticky record of the first NaN step ------------------------------------
namespace fr {
__global__ void loss_accumulate_kernel(const float* __restrict__ part, int n, float* __restrict__ acc) {
    bool bad = false;
    for (int q = 0; q < n; ++q) {
        const float x = part[q];
        acc[q] += x;
        bad |= x != x;
    }
    acc[3] += 1.f;
    if (bad && acc[4] == 0.f) acc[4] = acc[3];
}
}  // namespace fr

extern "C" int fr_loss_accumulate(const float* part, int32_t n, float* acc, void* stream) {
    FR_CHECK_ARG(part && acc && n >= 1 && n <= 3, "fr_loss_accumulate: 1..3 loss values, acc = float[8]");
    hipLaunchKernelGGL(fr::loss_accumulate_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, part, (int)n, acc);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
