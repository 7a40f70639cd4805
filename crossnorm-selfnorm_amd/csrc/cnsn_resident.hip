// Channel-resident strategy, the op alone: host entry points (shared logic in cnsn_resident_host.h).
#include "cnsn_resident_host.h"
#include "cnsn_env.h"

#include <atomic>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace cnsn {

namespace {
std::mutex g_flag_mu;
unsigned* g_host_flag = nullptr;
std::atomic<int> g_enabled{-1};  // -1: not decided yet (environment), 0: off, 1: on
}  // namespace

unsigned* resident_host_flag() {
    if (g_host_flag) return g_host_flag;
    std::lock_guard<std::mutex> lock(g_flag_mu);
    if (!g_host_flag) {
        void* p = nullptr;
        // 64 pinned bytes, once per process; fails harmlessly (stays NULL, retried later) inside a stream capture
        if (hipHostMalloc(&p, 64, hipHostMallocDefault) == hipSuccess && p) {
            memset(p, 0, 64);
            g_host_flag = (unsigned*)p;
        } else {
            (void)hipGetLastError();
        }
    }
    return g_host_flag;
}

int resident_timeouts() {
    const unsigned* f = g_host_flag;
    return f ? (int)*(volatile const unsigned*)f : 0;
}

namespace {
std::atomic<int> g_forgiven{0};  // time-outs that happened before the last cnsn_resident_rearm
std::atomic<int> g_rearms{0};
}  // namespace
bool resident_degraded() { return resident_timeouts() > g_forgiven.load(std::memory_order_relaxed); }
int resident_rearm() {
    g_forgiven.store(resident_timeouts(), std::memory_order_relaxed);
    return g_rearms.fetch_add(1, std::memory_order_relaxed) + 1;
}

bool resident_auto_enabled() {
    int e = g_enabled.load(std::memory_order_relaxed);
    if (e < 0) {
        const char* env = knob(K_RESIDENT);
        e = (env && env[0] == '0') ? 0 : 1;
        g_enabled.store(e, std::memory_order_relaxed);
    }
    return e == 1 && !resident_degraded();
}

void resident_set_enabled(bool on) { g_enabled.store(on ? 1 : 0, std::memory_order_relaxed); }

namespace {
std::atomic<int> g_wait_ms{0};  // cnsn_set_wait_ms: > 0 replaces the 5 s default; a CNSN_WAIT_MS knob in force still wins
}  // namespace
void resident_set_wait_ms(int ms) { g_wait_ms.store(ms > 0 ? ms : 0, std::memory_order_relaxed); }
long long resident_wait_ticks() {
    const char* wm = knob(K_WAIT_MS);  // (re-read by cnsn_reload_env: a knob set LATER than cnsn_set_wait_ms wins as well)
    if (wm && atoll(wm) > 0) return atoll(wm) * 100000ll;  // 100 MHz wall clock
    const int set = g_wait_ms.load(std::memory_order_relaxed);
    return set > 0 ? (long long)set * 100000ll : kWaitLimitTicks;
}

namespace {
std::atomic<int> g_headroom_cus{0};  // cnsn_set_headroom_cus; a CNSN_HEADROOM_CUS knob in force wins
}  // namespace
void resident_set_headroom_cus(int n) { g_headroom_cus.store(n > 0 ? n : 0, std::memory_order_relaxed); }
int resident_headroom_cus() {
    if (const char* hr = knob(K_HEADROOM_CUS)) return atoi(hr) > 0 ? atoi(hr) : 0;  // ("0" in the environment: none, whatever was set)
    return g_headroom_cus.load(std::memory_order_relaxed);
}

namespace {
std::mutex g_ctx_mu;
std::unordered_map<void*, unsigned> g_ctx_epoch;  // launches counted per context (host side)
}  // namespace

namespace {
struct PongState {
    size_t bytes = 0;          // context size the region addresses were derived from
    size_t clean[2] = {0, 0};  // leading bytes of each region known to hold 'empty'
    int next = 0;
};
std::unordered_map<void*, PongState> g_pong;  // (under g_ctx_mu)

// A launch that gave up leaves its context with the control word flipped (every later wait would drain at once), the granule
// regions half cleared and the barrier counter of the single-launch channels-last kernels short of arrivals.  Nobody uses the
// cluster kernels until somebody re-arms them (cnsn_resident_rearm); the first launch after that finds the count of time-outs
// changed and puts the control block back in order — on the stream, in launch order (all cluster launches are chained).
struct CtxHealth {
    int timeouts = 0;                    // resident_timeouts() when the context was last known to be in order
    size_t bytes = 0;                    // size the barrier block's position was derived from
    unsigned long long group_base = 0;   // arrivals the launches so far have left in every group counter of the barrier block
    unsigned long long bar_base = 0;     // barriers so far
};
std::unordered_map<void*, CtxHealth> g_health;  // (under g_ctx_mu)

CtxHealth& heal_context_locked(void* context, hipStream_t stream) {
    CtxHealth& h = g_health[context];
    const int now = resident_timeouts();
    if (h.timeouts != now) {
        if (hipMemsetAsync(context, 0, kCtlBytes, stream) != hipSuccess) (void)hipGetLastError();
        g_pong.erase(context);
        h.bytes = 0;  // (the barrier block is cleared by the next launch that uses it: resident_bar_area)
        h.timeouts = now;
    }
    return h;
}
}  // namespace

void resident_context_forget(void* context) {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    g_ctx_epoch[context] = 0;
    g_pong.erase(context);
    g_health[context] = CtxHealth{resident_timeouts(), 0, 0ull, 0ull};
}

BarArea resident_bar_area(const cnsn_problem_t& p, void* workspace_bar, hipStream_t stream, int grid, int barriers) {
    BarArea ba{(unsigned*)((char*)workspace_bar + kBarCtl * kBarLine), (char*)workspace_bar, 0ull, 0ull, true};
    if (!p.context || p.context_bytes < (uint64_t)kCtlBytes + kBarBlock + 2 * kPongRegion) return ba;
    if (const char* e = knob(K_CONTEXT))
        if (e[0] == '0') return ba;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return ba;  // a replay would repeat the bases: count from zero in the workspace instead
    }
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    if (g_ctx_epoch.find(p.context) == g_ctx_epoch.end()) return ba;  // never initialised through cnsn_context_init
    CtxHealth& h = heal_context_locked(p.context, stream);
    char* block = (char*)p.context + (size_t)p.context_bytes - 2 * kPongRegion - kBarBlock;
    if (h.bytes != (size_t)p.context_bytes) {  // first use, healed after a time-out, or the caller passes another extent
        if (hipMemsetAsync(block, 0, kBarBlock, stream) != hipSuccess) {
            (void)hipGetLastError();
            return ba;
        }
        h.bytes = (size_t)p.context_bytes;
        h.group_base = h.bar_base = 0;
    }
    ba.ctl = (unsigned*)p.context;
    ba.block = block;
    ba.group_base = h.group_base;
    ba.bar_base = h.bar_base;
    ba.need_fill = false;
    h.group_base += (unsigned long long)barriers * (unsigned long long)(grid / 8);
    h.bar_base += (unsigned long long)barriers;
    return ba;
}

bool resident_pong_acquire(const cnsn_problem_t& p, size_t fill_bytes, hipStream_t stream, PongArea* out) {
    // (+ 512: the scalar-path gather reads whole 256-byte groups, past the last granule)
    if (!p.context || fill_bytes + 512 > kPongRegion || (fill_bytes & 7) || p.context_bytes < 2 * kPongRegion + 4096) return false;
    if (const char* e = knob(K_PONG))
        if (e[0] == '0') return false;
    if (const char* e = knob(K_CONTEXT))
        if (e[0] == '0') return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return false;  // a replay would reuse the region chosen at capture time
    }
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    if (g_ctx_epoch.find(p.context) == g_ctx_epoch.end()) return false;  // never initialised through cnsn_context_init
    (void)heal_context_locked(p.context, stream);
    PongState& st = g_pong[p.context];
    if (st.bytes != (size_t)p.context_bytes) st = PongState{(size_t)p.context_bytes, {0, 0}, 0};
    const int r = st.next;
    char* end = (char*)p.context + (size_t)p.context_bytes;
    char* region[2] = {end - 2 * kPongRegion, end - kPongRegion};
    out->base = region[r];
    out->clear = (unsigned long long*)region[r ^ 1];
    out->clear_qwords = (unsigned)(fill_bytes / 8);
    out->need_fill = st.clean[r] < fill_bytes;
    st.clean[r] = 0;  // dirty from now on, launched or not (the fill in front of an abandoned launch leaves it clean: unused)
    return true;
}

void resident_pong_commit(const cnsn_problem_t& p, size_t fill_bytes) {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    auto it = g_pong.find(p.context);
    if (it == g_pong.end()) return;
    PongState& st = it->second;
    st.clean[st.next ^ 1] = fill_bytes;  // this launch's workgroups store 'empty' over that much of the other region
    st.next ^= 1;
}

ExchangeArea resident_exchange_area(const cnsn_problem_t& p, size_t tagged_bytes, void* workspace, hipStream_t stream,
                                    bool prefer_context) {
    ExchangeArea ea{workspace, 0u};
    if (!p.context || p.context_bytes < tagged_bytes + kBarBlock + 2 * kPongRegion) return ea;  // (the barrier block and the two regions at the end are not for tagged granules)
    if (const char* e = knob(K_CONTEXT)) {
        if (e[0] == '0') return ea;
    } else if (!prefer_context && ((size_t)p.N * p.C * p.H * p.W * elem_bytes(p.dtype) >= ((size_t)64 << 20) ||
                                   (knob(K_PONG) && knob(K_PONG)[0] == '2'))) {  // (CNSN_PONG=2, tests: small tensors too)
        // Large tensors exchange through the workspace: a tagged granule carries ONE float per 8 bytes, an untagged one
        // two, and at this size the gather of a channel's granules (4 KB tagged at N = 256, 12 KB with crop boxes) costs
        // more than the fill launch the context saves — north-star shape 0.849 -> 0.839 ms per step, with crop boxes
        // 1.114 -> 1.057.  (CNSN_CONTEXT=1 keeps the context for every size.)
        return ea;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return ea;  // a captured launch would replay its number: exchange through the workspace instead
    }
    unsigned epoch;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        auto it = g_ctx_epoch.find(p.context);
        if (it == g_ctx_epoch.end()) return ea;  // never initialised through cnsn_context_init: not trusted
        (void)heal_context_locked(p.context, stream);
        if (const char* e = knob(K_EPOCH_START))  // (tests: start close to the wrap-around)
            if (it->second == 0) it->second = (unsigned)strtoul(e, nullptr, 0);
        epoch = ++it->second;
        if (epoch == 0) {  // wrapped: every tag in the context is stale-but-plausible now — clear it once, in order
            if (hipMemsetAsync(p.context, 0, (size_t)p.context_bytes, stream) != hipSuccess) {
                (void)hipGetLastError();
                it->second = 0xffffffffu;  // try again next time
                return ea;
            }
            epoch = it->second = 1;
            g_pong.erase(p.context);  // (the clear zeroed the granule regions too: not 'empty' any more)
            g_health[p.context].bytes = 0;  // (... and the barrier block: its bases start again)
        }
    }
    ea.base = p.context;
    ea.epoch = epoch;
    return ea;
}

namespace {
struct ChainState {
    std::mutex mu;
    hipEvent_t ev = nullptr;
    hipStream_t last = nullptr;
    bool valid = false;
};
ChainState g_chain[16];
}  // namespace

// Round 4: the completion event is recorded WHEN A LAUNCH ARRIVES ON ANOTHER STREAM (on the stream of the previous cluster
// launch: everything queued there so far, that launch included), not behind every launch.  An event record is a packet of
// its own on the queue and costs the GPU about 4 us between two dependent launches (rocprofv3 timeline of the headline
// step, profiles/r04_launches_per_step.md: 10.7 / 14.9 us gaps with two / three records in them); a training loop stays
// on one stream and never pays it now.
ResidentChain::ResidentChain(hipStream_t stream) : stream_(stream), dev_(0), active_(false) {
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 16) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return;
    }
    ChainState& c = g_chain[dev_];
    c.mu.lock();
    active_ = true;
    if (c.valid && c.last != stream) {
        hipStreamCaptureStatus ls = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c.last, &ls) != hipSuccess) {
            // the stream of the previous cluster launch no longer exists: nothing to record on — drain the device once
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
        } else if (ls == hipStreamCaptureStatusNone) {
            if (!c.ev && hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess) c.ev = nullptr;
            if (c.ev && hipEventRecord(c.ev, c.last) == hipSuccess) {
                (void)hipStreamWaitEvent(stream, c.ev, 0);
            } else {
                (void)hipGetLastError();
                (void)hipDeviceSynchronize();
            }
        }  // (that stream is under capture now: its earlier, un-captured work was drained when the capture began)
    }
}

ResidentChain::~ResidentChain() {
    if (!active_) return;
    ChainState& c = g_chain[dev_];
    c.last = stream_;
    c.valid = true;
    c.mu.unlock();
}

ResPlan resident_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, bool backward) {
    return reshost::plan_impl(p, boxed, has_chan_perm, backward, false);
}

int resident_forward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* x,
                     const int64_t* perm, GateDev g, GateDev f, void* y, double* saved, void* workspace,
                     hipStream_t stream) {
    return reshost::forward_impl<false>(p, cb, sb, boxed, mid, x, nullptr, 0, perm, g, f, y, saved, workspace, stream);
}

size_t resident_workspace_bytes(const cnsn_problem_t& p, bool boxed) {
    return kCtlBytes + (size_t)p.N * p.C * (boxed ? 6 : 2) * 8 + 256;  // (+ 256: whole 256-byte groups are read)
}

}  // namespace cnsn
