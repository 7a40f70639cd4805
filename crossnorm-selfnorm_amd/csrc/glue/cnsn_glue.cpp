// C++ autograd glue between PyTorch-ROCm and the C ABI of libcnsn_hip.so.
//
// torch here is plumbing only — device memory (caching allocator), the current HIP stream, autograd
// book-keeping: every forward/backward is ONE call through include/cnsn_hip.h.  This file does the same
// job as functional.py's ctypes path (which stays as the alternative when this module has not been
// built); it exists because the Python glue costs ~0.1 ms of host time per call, which is what bounds a
// network made of many small CNSN sites (WideResNet-40-2: 18 sites at 32x32 and below).
#include <torch/extension.h>

#include <ATen/hip/MemPool.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <torch/csrc/cuda/CUDAPluggableAllocator.h>

#include <array>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/cnsn_hip.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

int dtype_code(const Tensor& x) {
    switch (x.scalar_type()) {
        case at::kFloat: return CNSN_F32;
        case at::kBFloat16: return CNSN_BF16;
        case at::kHalf: return CNSN_F16;
        default: TORCH_CHECK_TYPE(false, "cnsn: dtype ", x.scalar_type(), " not supported (float32, bfloat16, float16)");
    }
}

void check_status(int st, const char* what) {
    if (st == 0) return;
    if (st == CNSN_E_BATCH) {  // the exception type nn.BatchNorm1d raises
        TORCH_CHECK_VALUE(false, what, ": ", cnsn_status_string(st));
    }
    TORCH_CHECK(false, what, " failed with status ", st, ": ", cnsn_status_string(st));
}

// contiguous AND 16-byte aligned: `.contiguous()` keeps a contiguous view with a storage offset (x[1:], a
// torch.split chunk) as it is; the kernels' 16-byte vector accesses need an aligned base, so such a view is copied
Tensor dense(const Tensor& t) {
    Tensor d = t.contiguous();
    if ((reinterpret_cast<uintptr_t>(d.data_ptr()) & 15u) != 0) d = d.clone(at::MemoryFormat::Contiguous);
    return d;
}

Tensor f32c(const Tensor& t) {
    Tensor d = t.detach();
    if (d.scalar_type() != at::kFloat) d = d.to(at::kFloat);
    return d.contiguous();
}

// ---- output arena (include/cnsn_hip.h, "output arena"): y / dx / z of at least `g_arena_min` bytes come from a torch MemPool
// whose blocks the LIBRARY creates (cnsn_arena_map: address ranges mapped from physical allocations of its own, a new block of
// 384 MiB or more the fastest of several candidates timed when it is created — where a block lies physically decides how fast
// the single-touch launches write it, profiles/r05_arena.md).  Everything else about those blocks is the caching allocator's:
// it caches and splits them, `torch.cuda.memory_allocated` counts them, `Tensor.record_stream` orders their re-use across
// streams, its out-of-memory path releases them, and (`use_on_oom`) an allocation ANYWHERE in the process that would otherwise
// fail may use the pool's free blocks.  Round 5 handed out at::from_blob tensors over a cache of the library's own, which
// torch could neither see, trim nor order (review of round 5, both findings).  Outputs below the threshold, and every output
// while the stream is being captured into a graph, come from torch's default pool.
// CNSN_ARENA=0 switches it off, CNSN_ARENA_MIN_MB moves the threshold (default 32); `arena_config` does both at run time.
std::atomic<int64_t> g_arena_min{-2};  // -2: not read yet, -1: off

int64_t arena_min_bytes() {
    int64_t v = g_arena_min.load(std::memory_order_relaxed);
    if (v == -2) {
        const char* on = getenv("CNSN_ARENA");
        const char* mb = getenv("CNSN_ARENA_MIN_MB");
        v = (on && on[0] == '0') ? -1 : (int64_t)((mb && atoll(mb) > 0) ? atoll(mb) : 32) << 20;
        g_arena_min.store(v, std::memory_order_relaxed);
    }
    return v;
}

namespace hca = c10::hip::HIPCachingAllocator;

struct ArenaPools {
    std::mutex mu;
    std::shared_ptr<hca::HIPAllocator> allocator;
    std::unordered_map<int, std::shared_ptr<at::cuda::MemPool>> by_device;
    bool unusable = false;  // this torch build refused the pool once: torch's default pool from then on
};
ArenaPools& arena_pools() {
    static ArenaPools* p = new ArenaPools;  // (leaked on purpose: tensors outlive static destructors)
    return *p;
}

// the pool of `device` (created on first use with that device current), or nullptr
std::shared_ptr<at::cuda::MemPool> arena_pool(int device) {
    ArenaPools& ps = arena_pools();
    std::lock_guard<std::mutex> lock(ps.mu);
    if (ps.unusable) return nullptr;
    auto it = ps.by_device.find(device);
    if (it != ps.by_device.end()) return it->second;
    try {
        if (!ps.allocator)
            ps.allocator = torch::cuda::CUDAPluggableAllocator::createCustomAllocator(
                [](size_t bytes, int dev, hipStream_t stream) { return cnsn_arena_map(bytes, dev, (void*)stream); },
                [](void* ptr, size_t bytes, int dev, hipStream_t stream) { cnsn_arena_unmap(ptr, bytes, dev, (void*)stream); });
        c10::DeviceGuard guard(c10::Device(c10::DeviceType::CUDA, (c10::DeviceIndex)device));
        std::shared_ptr<at::cuda::MemPool> pool(new at::cuda::MemPool(ps.allocator.get(), /*is_user_created=*/true, /*use_on_oom=*/true));
        ps.by_device.emplace(device, pool);
        return pool;
    } catch (const c10::Error&) {
        ps.unusable = true;
        return nullptr;
    }
}

// allocations of THIS thread on `device` go to the arena's pool while one of these is alive (torch.cuda.use_mem_pool's steps)
struct ToArenaPool {
    c10::DeviceIndex dev;
    c10::MempoolId_t id;
    std::shared_ptr<at::cuda::MemPool> pool;  // (a trim may retire the pool meanwhile: the last holder destroys it)
    ToArenaPool(int device, std::shared_ptr<at::cuda::MemPool> p) : dev((c10::DeviceIndex)device), id(p->id()), pool(std::move(p)) {
        const auto tid = std::this_thread::get_id();
        hca::beginAllocateToPool(dev, id, [tid](hipStream_t) { return std::this_thread::get_id() == tid; });
    }
    ~ToArenaPool() {
        hca::endAllocateToPool(dev, id);
        hca::releasePool(dev, id);
        pool.reset();  // (after the pool's use count is back to what ~MemPool expects)
    }
};

Tensor arena_empty(at::IntArrayRef sizes, at::IntArrayRef strides, const at::TensorOptions& opt, const at::Device& dev, int64_t /*nbytes*/) {
    hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return Tensor();
    }
    std::shared_ptr<at::cuda::MemPool> pool = arena_pool((int)dev.index());
    if (!pool) return Tensor();
    ToArenaPool scope((int)dev.index(), std::move(pool));
    // (out of memory: the caching allocator has by now released its own caches AND this pool's; the error is the one the
    // reference's plain allocation would raise)
    return at::empty_strided(sizes, strides, opt.device(dev));
}

// Give the free cached blocks of the arena's pool on `device` (< 0: every device) back to the driver.  The caching allocator
// releases the cached blocks of a user pool only once nobody holds the pool (`emptyCache(id)` walks the FREEABLE pools), so a
// trim RETIRES the pool: ~MemPool drops the last reference and empties its cache, tensors still alive keep their blocks (which
// go back to the retired pool and are released by the next `torch.cuda.empty_cache()` or out-of-memory retry), and the next
// output gets a fresh pool.
void arena_pool_trim(int device) {
    ArenaPools& ps = arena_pools();
    std::vector<std::shared_ptr<at::cuda::MemPool>> retired;
    {
        std::lock_guard<std::mutex> lock(ps.mu);
        for (auto it = ps.by_device.begin(); it != ps.by_device.end();) {
            if (device < 0 || it->first == device) {
                retired.push_back(std::move(it->second));
                it = ps.by_device.erase(it);
            } else {
                ++it;
            }
        }
    }
    retired.clear();  // (~MemPool here, or in the thread that is allocating from it right now when its scope ends)
}

// `x` is dense (contiguous in NCHW or channels-last order, aligned): a fresh tensor of its shape, type AND memory order for an
// output of the op
Tensor out_like(const Tensor& x) {
    const int64_t nbytes = x.numel() * (int64_t)x.element_size();
    const int64_t min = arena_min_bytes();
    if (min >= 0 && nbytes >= min) {
        Tensor t = arena_empty(x.sizes(), x.strides(), at::TensorOptions().dtype(x.scalar_type()).device(x.device()), x.device(),
                               nbytes);
        if (t.defined()) return t;
    }
    return at::empty_like(x);
}

// Channels-last calls the library computes where the tensor lies (cnsn_problem_t.layout = CNSN_LAYOUT_NHWC, include/cnsn_hip.h):
// strictly channels-last strides, un-boxed, no channel permutation, a channel count that is a whole number of 16-byte vectors.
// Everything else is copied to NCHW like the reference's `.contiguous()` (models/cnsn.py:14).  CNSN_NHWC=0 switches it off.
bool nhwc_call(const Tensor& x, const int64_t* cbox, const int64_t* sbox, bool has_chan) {
    static const bool on = [] {
        const char* e = getenv("CNSN_NHWC");
        return !(e && e[0] == '0');
    }();
    if (!on || x.dim() != 4 || has_chan || cbox[0] >= 0 || sbox[0] >= 0) return false;
    if (x.is_contiguous() || !x.is_contiguous(at::MemoryFormat::ChannelsLast)) return false;
    const int64_t vec = 16 / (int64_t)x.element_size();
    return x.size(1) % vec == 0 && x.size(2) * x.size(3) >= 2 && (reinterpret_cast<uintptr_t>(x.data_ptr()) & 15u) == 0;
}
Tensor dense_cl(const Tensor& t) {  // channels-last contiguous AND 16-byte aligned
    Tensor d = t.contiguous(at::MemoryFormat::ChannelsLast);
    if ((reinterpret_cast<uintptr_t>(d.data_ptr()) & 15u) != 0) d = d.clone(at::MemoryFormat::ChannelsLast);
    return d;
}

// host -> device copy of the (tiny) permutation through a ring of pinned staging buffers, so that the
// launch thread never waits for the stream (the reference's `torch.randperm(N).to(device)` does)
Tensor perm_to_device(const Tensor& idx, const at::Device& dev) {
    if (idx.is_cuda()) return idx.to(dev, at::kLong).contiguous();
    struct Slot {
        Tensor buf;
        hipEvent_t ev = nullptr;
    };
    static std::mutex mu;
    static std::unordered_map<int64_t, std::pair<int, std::array<Slot, 8>>> rings;
    const int64_t n = idx.numel();
    std::lock_guard<std::mutex> lock(mu);
    auto& ring = rings[(int64_t)dev.index() * (1ll << 40) + n];
    Slot& s = ring.second[ring.first++ % 8];
    if (!s.buf.defined()) {
        s.buf = at::empty({n}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
        TORCH_CHECK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) == hipSuccess, "cnsn: hipEventCreate");
    } else {
        hipEventSynchronize(s.ev);  // the copy that last used this slot has drained
    }
    s.buf.copy_(idx.reshape({-1}));
    Tensor out = s.buf.to(dev, /*non_blocking=*/true);
    hipEventRecord(s.ev, c10::hip::getCurrentHIPStream(dev.index()).stream());
    return out;
}

struct Config {
    int64_t cn_active, cbox[4], sbox[4];
    double lam;
    int64_t sn_active, sn_two, sn_training;
    double eps_cn, eps_sn, eps_bn, momentum;
    int64_t strategy;
    int64_t add_mode, relu;  // residual-block epilogue (cnsn_epilogue_t)
    int64_t perm_inline;     // the caller's plan: the permutation can ride in the launch arguments (cnsn_problem_t.perm_host)
};

Config parse_config(const std::vector<int64_t>& cfg, const std::vector<double>& fcfg) {
    TORCH_CHECK(cfg.size() == 17 && fcfg.size() == 5, "cnsn glue: bad config vectors");
    Config c{};
    c.cn_active = cfg[0];
    for (int i = 0; i < 4; ++i) {
        c.cbox[i] = cfg[1 + i];
        c.sbox[i] = cfg[5 + i];
    }
    c.sn_active = cfg[9];
    c.sn_two = cfg[10];
    c.sn_training = cfg[11];
    c.strategy = cfg[12];
    c.add_mode = cfg[14];
    c.relu = cfg[15];
    c.perm_inline = cfg[16];
    c.lam = fcfg[0];
    c.eps_cn = fcfg[1];
    c.eps_sn = fcfg[2];
    c.eps_bn = fcfg[3];
    c.momentum = fcfg[4];
    return c;
}

cnsn_epilogue_t make_epilogue(const Config& c, const Tensor& addend) {
    cnsn_epilogue_t e{};
    e.struct_bytes = (int32_t)sizeof(cnsn_epilogue_t);
    e.add_mode = (int32_t)c.add_mode;
    e.relu = (int32_t)c.relu;
    e.addend = addend.defined() ? addend.data_ptr() : nullptr;
    return e;
}

cnsn_problem_t make_problem(const Tensor& x, const Config& c) {
    cnsn_problem_t p{};
    p.struct_bytes = (int32_t)sizeof(cnsn_problem_t);
    p.dtype = dtype_code(x);
    p.N = (int32_t)x.size(0);
    p.C = (int32_t)x.size(1);
    p.H = (int32_t)x.size(2);
    p.W = (int32_t)x.size(3);
    p.cn_active = (int32_t)c.cn_active;
    for (int i = 0; i < 4; ++i) {
        p.content_box[i] = (int32_t)c.cbox[i];
        p.style_box[i] = (int32_t)c.sbox[i];
    }
    p.lam = (float)c.lam;
    p.eps_cn = (float)c.eps_cn;
    p.sn_active = (int32_t)c.sn_active;
    p.sn_two = (int32_t)c.sn_two;
    p.sn_training = (int32_t)c.sn_training;
    p.eps_sn = (float)c.eps_sn;
    p.eps_bn = (float)c.eps_bn;
    p.momentum = (float)c.momentum;
    p.strategy = (int32_t)c.strategy;
    return p;
}

// The persistent exchange context of the cluster-resident kernels (cnsn_context_init, include/cnsn_hip.h): one buffer
// per device, grown when a larger problem shows up; left out while the stream is being captured into a graph.
void attach_context(cnsn_problem_t& p, const at::Device& dev, hipStream_t stream) {
    p.context = nullptr;
    p.context_bytes = 0;
    const size_t need = cnsn_context_bytes(&p);
    if (need == 0) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return;
    }
    static std::mutex mu;
    static std::unordered_map<int, Tensor> ctx;
    static std::vector<Tensor> retired;  // outgrown buffers stay alive: launches queued on OTHER streams may still exchange through them
    std::lock_guard<std::mutex> lock(mu);
    Tensor& have = ctx[(int)dev.index()];
    if (!have.defined() || (size_t)have.numel() < need) {
        const size_t size = std::max(need, have.defined() ? 2 * (size_t)have.numel() : (size_t)(4u << 20));
        Tensor buf = at::empty({(int64_t)size}, at::TensorOptions().dtype(at::kByte).device(dev));
        check_status(cnsn_context_init(buf.data_ptr(), size, (void*)stream), "cnsn_context_init");
        TORCH_CHECK(hipStreamSynchronize(stream) == hipSuccess, "cnsn: hipStreamSynchronize");   // once per buffer
        if (have.defined()) retired.push_back(have);
        have = buf;
    }
    p.context = have.data_ptr();
    p.context_bytes = (uint64_t)have.numel();
}

struct GateTensors {  // float32 contiguous views/copies + where running stats must be copied back to
    Tensor w, gamma, beta, rm, rv, rm_src, rv_src;
    bool direct = true;
    cnsn_gate_t c{};
    // counter: nn.BatchNorm1d.num_batches_tracked when this call has to count (the forward kernel adds 1), else undefined
    void init(const Tensor& w_, const Tensor& g_, const Tensor& b_, const Tensor& rm_, const Tensor& rv_,
              const c10::optional<Tensor>& counter = c10::nullopt) {
        w = f32c(w_);
        gamma = f32c(g_);
        beta = f32c(b_);
        direct = rm_.scalar_type() == at::kFloat && rm_.is_contiguous() && rv_.scalar_type() == at::kFloat &&
                 rv_.is_contiguous();
        rm_src = rm_;
        rv_src = rv_;
        rm = direct ? rm_.detach() : f32c(rm_);
        rv = direct ? rv_.detach() : f32c(rv_);
        c.fc_weight = w.data_ptr<float>();
        c.bn_weight = gamma.data_ptr<float>();
        c.bn_bias = beta.data_ptr<float>();
        c.running_mean = rm.data_ptr<float>();
        c.running_var = rv.data_ptr<float>();
        c.num_batches_tracked = nullptr;
        if (counter.has_value() && counter->defined()) {
            if (counter->is_cuda() && counter->scalar_type() == at::kLong && counter->numel() == 1)
                c.num_batches_tracked = counter->data_ptr<int64_t>();
            else
                counter->add_(1);  // a counter the kernel cannot reach: counted here, as nn.BatchNorm1d.forward does
        }
    }
    void write_back() {
        if (!direct) {
            rm_src.copy_(rm);
            rv_src.copy_(rv);
        }
    }
};

// CNSN_KEEP_SUM=0: never keep x + addend for the backward (A/B knob)
bool keep_sum_enabled() {
    static const bool on = [] {
        const char* e = std::getenv("CNSN_KEEP_SUM");
        return !(e && e[0] == '0');
    }();
    return on;
}

class FusedCNSN : public torch::autograd::Function<FusedCNSN> {
   public:
    // cfg: [cn_active, cb0..3, sb0..3, sn_active, sn_two, sn_training, strategy, need_backward, add_mode, relu]
    // fcfg: [lam, eps_cn, eps_sn, eps_bn, momentum]
    static Tensor forward(AutogradContext* ctx, const Tensor& x_in, std::vector<int64_t> cfg, std::vector<double> fcfg,
                          const c10::optional<Tensor>& perm_in, const c10::optional<Tensor>& chan_in,
                          const c10::optional<Tensor>& g_w, const c10::optional<Tensor>& g_gamma,
                          const c10::optional<Tensor>& g_beta, const c10::optional<Tensor>& g_rm,
                          const c10::optional<Tensor>& g_rv, const c10::optional<Tensor>& f_w,
                          const c10::optional<Tensor>& f_gamma, const c10::optional<Tensor>& f_beta,
                          const c10::optional<Tensor>& f_rm, const c10::optional<Tensor>& f_rv,
                          const c10::optional<Tensor>& addend_in, const c10::optional<Tensor>& g_nbt,
                          const c10::optional<Tensor>& f_nbt) {
        TORCH_CHECK(x_in.is_cuda(), "cnsn_forward: got a ", x_in.device().type(),
                    " tensor. This implementation runs on MI355X HIP device tensors only; there is no CPU path.");
        TORCH_CHECK(x_in.dim() == 4, "expected an (N, C, H, W) tensor");
        const Config c = parse_config(cfg, fcfg);
        // the library sizes grids / orders its persistent launches for the CURRENT device and the stream below is the
        // tensor's device's: make that device current (a model on cuda:1 must not need set_device(1))
        const c10::DeviceGuard device_guard(x_in.device());

        const bool nhwc = nhwc_call(x_in, c.cbox, c.sbox, c.cn_active && chan_in.has_value());
        const Tensor x = nhwc ? x_in : dense(x_in);  // reference cnsn.py:14 (a channels-last call is computed where it lies)
        Tensor addend;
        if (c.add_mode != CNSN_ADD_NONE) {
            TORCH_CHECK(addend_in.has_value() && addend_in->is_cuda() && addend_in->sizes() == x.sizes() &&
                            addend_in->scalar_type() == x.scalar_type(),
                        "cnsn_forward: the addend must be a device tensor of x's shape and dtype");
            addend = nhwc ? dense_cl(*addend_in) : dense(*addend_in);
        }
        cnsn_problem_t prob = make_problem(x, c);
        prob.layout = nhwc ? CNSN_LAYOUT_NHWC : CNSN_LAYOUT_NCHW;
        cnsn_epilogue_t epi = make_epilogue(c, addend);
        const bool has_epi = c.add_mode != CNSN_ADD_NONE || c.relu;
        const at::Device dev = x.device();
        attach_context(prob, dev, c10::hip::getCurrentHIPStream(dev.index()).stream());
        Tensor perm, chan, perm_host;
        if (c.cn_active) {
            TORCH_CHECK(perm_in.has_value(), "cnsn_forward: CrossNorm needs the batch permutation");
            const bool inline_ok = !nhwc && c.perm_inline && !chan_in.has_value() && !perm_in->is_cuda() &&
                                   perm_in->scalar_type() == at::kLong && perm_in->is_contiguous() &&
                                   perm_in->numel() <= CNSN_PERM_INLINE_MAX && perm_in->numel() == x.size(0);
            if (inline_ok) {
                // rides in the launch arguments: no host-to-device copy.  A SNAPSHOT when a backward will read it again
                // (<= 8 KB): the caller may reuse its index buffer between the two (`randperm(out=buf)`)
                perm_host = cfg[13] != 0 ? perm_in->clone() : *perm_in;
                prob.perm_host = perm_host.data_ptr<int64_t>();
            } else {
                perm = perm_to_device(*perm_in, dev);
                if (chan_in.has_value()) chan = perm_to_device(*chan_in, dev);
            }
        }
        GateTensors gg, gf;
        const bool two = c.sn_active && c.sn_two;
        if (c.sn_active) gg.init(*g_w, *g_gamma, *g_beta, *g_rm, *g_rv, c.sn_training ? g_nbt : c10::nullopt);
        if (two) gf.init(*f_w, *f_gamma, *f_beta, *f_rm, *f_rv, c.sn_training ? f_nbt : c10::nullopt);

        Tensor y = out_like(x);
        const bool need_bwd = cfg[13] != 0;  // decided by the caller (grad mode on and something requires grad)
        // A PRE add in front of a channels-last call (two tensor passes each way): the forward KEEPS X = x + addend and the
        // backward reads that one tensor instead of two, twice (cnsn_epilogue_t.sum_out, ABI 8).  X is what the reference's
        // in-place `out += identity` leaves (resnet_cnsn.py:117) and what its autograd saves; x and the addend are not saved.
        Tensor xsum;
        if (need_bwd && nhwc && c.add_mode == CNSN_ADD_PRE && keep_sum_enabled() && cnsn_keeps_sum(&prob, &epi) == 1) {
            xsum = out_like(x);
            epi.sum_out = xsum.data_ptr();
        }
        const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
        const size_t ws_bytes = cnsn_workspace_bytes(&prob);
        Tensor saved;
        if (need_bwd && ws_bytes > 0) saved = at::empty({(int64_t)cnsn_saved_floats(&prob)}, fopt);
        Tensor ws = at::empty({(int64_t)(ws_bytes / 4) + 1}, fopt);
        hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        auto launch = [&]() {
            return cnsn_forward_fused(&prob, has_epi ? &epi : nullptr, x.data_ptr(),
                                      perm.defined() ? perm.data_ptr<int64_t>() : nullptr,
                                      chan.defined() ? chan.data_ptr<int64_t>() : nullptr, c.sn_active ? &gg.c : nullptr,
                                      two ? &gf.c : nullptr, y.data_ptr(), saved.defined() ? saved.data_ptr<float>() : nullptr,
                                      ws.data_ptr(), ws_bytes, (void*)stream);
        };
        int st = launch();
        if (st == CNSN_E_UNSUPPORTED && perm_host.defined()) {  // (the plan changed under the caller: upload, call again)
            perm = perm_to_device(perm_host, dev);
            perm_host = Tensor();
            prob.perm_host = nullptr;
            st = launch();
        }
        check_status(st, "cnsn_forward");
        if (c.sn_active && c.sn_training) {
            gg.write_back();
            if (two) gf.write_back();
        }
        if (need_bwd) {
            ctx->saved_data["kept_sum"] = xsum.defined();
            if (xsum.defined()) cfg[14] = CNSN_ADD_NONE;  // the backward is the backward of the op WITHOUT the add, on X
            ctx->saved_data["cfg"] = cfg;
            ctx->saved_data["fcfg"] = fcfg;
            ctx->saved_data["pd"] = std::vector<int64_t>{
                c.sn_active ? (int64_t)g_w->scalar_type() : -1, c.sn_active ? (int64_t)g_gamma->scalar_type() : -1,
                c.sn_active ? (int64_t)g_beta->scalar_type() : -1, two ? (int64_t)f_w->scalar_type() : -1,
                two ? (int64_t)f_gamma->scalar_type() : -1, two ? (int64_t)f_beta->scalar_type() : -1};
            Tensor none;
            ctx->saved_data["perm_host"] = perm_host.defined() ? c10::IValue(perm_host) : c10::IValue();
            ctx->save_for_backward({xsum.defined() ? xsum : x, saved, perm.defined() ? perm : none, chan.defined() ? chan : none,
                                    c.sn_active ? gg.w : none, c.sn_active ? gg.gamma : none, c.sn_active ? gg.beta : none,
                                    c.sn_active ? gg.rm : none, c.sn_active ? gg.rv : none, two ? gf.w : none,
                                    two ? gf.gamma : none, two ? gf.beta : none, two ? gf.rm : none, two ? gf.rv : none,
                                    (addend.defined() && !xsum.defined()) ? addend : none});
        }
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto sv = ctx->get_saved_variables();
        const Tensor &x = sv[0], &saved = sv[1], &chan = sv[3], &addend = sv[14];
        Tensor perm = sv[2];
        Tensor perm_host;
        if (ctx->saved_data.count("perm_host") && ctx->saved_data["perm_host"].isTensor())
            perm_host = ctx->saved_data["perm_host"].toTensor();
        const auto cfg = ctx->saved_data["cfg"].toIntVector();
        const auto fcfg = ctx->saved_data["fcfg"].toDoubleVector();
        const auto pd = ctx->saved_data["pd"].toIntVector();
        const Config c = parse_config(cfg, fcfg);
        cnsn_problem_t prob = make_problem(x, c);
        const bool nhwc = nhwc_call(x, c.cbox, c.sbox, chan.defined());  // (the saved x: the forward's decision again)
        prob.layout = nhwc ? CNSN_LAYOUT_NHWC : CNSN_LAYOUT_NCHW;
        if (perm_host.defined()) prob.perm_host = perm_host.data_ptr<int64_t>();
        const bool two = c.sn_active && c.sn_two;
        const at::Device dev = x.device();
        const c10::DeviceGuard device_guard(dev);
        attach_context(prob, dev, c10::hip::getCurrentHIPStream(dev.index()).stream());

        Tensor gy = grads[0];
        if (gy.scalar_type() != x.scalar_type()) gy = gy.to(x.scalar_type());
        gy = nhwc ? dense_cl(gy) : dense(gy);
        Tensor dx = out_like(x);
        const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
        const size_t ws_bytes = cnsn_workspace_bytes(&prob);
        Tensor ws = at::empty({(int64_t)(ws_bytes / 4) + 1}, fopt);
        const int64_t Cn = x.size(1);

        cnsn_gate_t gg{}, gf{};
        cnsn_gate_grad_t dgg{}, dgf{};
        Tensor flat_g, flat_f;
        auto fill_gate = [&](cnsn_gate_t& g, int base) {
            g.fc_weight = sv[base].data_ptr<float>();
            g.bn_weight = sv[base + 1].data_ptr<float>();
            g.bn_bias = sv[base + 2].data_ptr<float>();
            g.running_mean = sv[base + 3].data_ptr<float>();
            g.running_var = sv[base + 4].data_ptr<float>();
            g.num_batches_tracked = nullptr;
        };
        auto make_grads = [&](Tensor& flat, cnsn_gate_grad_t& d) {  // one allocation: dw (C,1,2) | dgamma (C) | dbeta (C)
            flat = at::empty({4 * Cn}, fopt);
            d.d_fc_weight = flat.data_ptr<float>();
            d.d_bn_weight = flat.data_ptr<float>() + 2 * Cn;
            d.d_bn_bias = flat.data_ptr<float>() + 3 * Cn;
        };
        if (c.sn_active) {
            fill_gate(gg, 4);
            make_grads(flat_g, dgg);
            if (two) {
                fill_gate(gf, 9);
                make_grads(flat_f, dgf);
            }
        }
        hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        const cnsn_epilogue_t epi = make_epilogue(c, addend);
        const bool has_epi = c.add_mode != CNSN_ADD_NONE || c.relu;
        Tensor d_add;  // gradient of the addend: dx itself (PRE), grad_y behind the ReLU mask (POST)
        if (c.add_mode == CNSN_ADD_POST) d_add = c.relu ? out_like(x) : gy;
        auto launch = [&]() {
            return cnsn_backward_fused(
                &prob, has_epi ? &epi : nullptr, gy.data_ptr(), x.data_ptr(), perm.defined() ? perm.data_ptr<int64_t>() : nullptr,
                chan.defined() ? chan.data_ptr<int64_t>() : nullptr, c.sn_active ? &gg : nullptr, two ? &gf : nullptr,
                saved.data_ptr<float>(), dx.data_ptr(), (c.add_mode == CNSN_ADD_POST && c.relu) ? d_add.data_ptr() : nullptr,
                c.sn_active ? &dgg : nullptr, two ? &dgf : nullptr, ws.data_ptr(), ws_bytes, (void*)stream);
        };
        int st = launch();
        if (st == CNSN_E_UNSUPPORTED && perm_host.defined()) {  // (another strategy by now: it wants the device array)
            perm = perm_to_device(perm_host, dev);
            prob.perm_host = nullptr;
            st = launch();
        }
        if (c.add_mode == CNSN_ADD_PRE || ctx->saved_data["kept_sum"].toBool()) d_add = dx;
        check_status(st, "cnsn_backward");

        auto cast = [](const Tensor& t, int64_t code) {
            return (code >= 0 && (int64_t)t.scalar_type() != code) ? t.to((at::ScalarType)code) : t;
        };
        Tensor none;
        variable_list out(18, none);
        out[0] = dx;
        out[15] = d_add;
        if (c.sn_active) {
            out[5] = cast(flat_g.narrow(0, 0, 2 * Cn).view({Cn, 1, 2}), pd[0]);
            out[6] = cast(flat_g.narrow(0, 2 * Cn, Cn), pd[1]);
            out[7] = cast(flat_g.narrow(0, 3 * Cn, Cn), pd[2]);
            if (two) {
                out[10] = cast(flat_f.narrow(0, 0, 2 * Cn).view({Cn, 1, 2}), pd[3]);
                out[11] = cast(flat_f.narrow(0, 2 * Cn, Cn), pd[4]);
                out[12] = cast(flat_f.narrow(0, 3 * Cn, Cn), pd[5]);
            }
        }
        return out;
    }
};


// (y, z) = (SelfNorm(x [+ addend]), relu(BatchNorm2d(y))) — cnsn_forward_bnrelu / cnsn_backward_bnrelu: the end of one
// WideResNet block and the start of the next in one launch per direction (wideresnet_cnsn.py:93-96, :76-77).
class FusedCNSNTail : public torch::autograd::Function<FusedCNSNTail> {
   public:
    // tcfg: [want_y, bn_training]; tf: [bn_eps, bn_momentum]
    static variable_list forward(AutogradContext* ctx, const Tensor& x_in, std::vector<int64_t> cfg, std::vector<double> fcfg,
                                 const Tensor& g_w, const Tensor& g_gamma, const Tensor& g_beta, const Tensor& g_rm,
                                 const Tensor& g_rv, const c10::optional<Tensor>& addend_in, const Tensor& bn_w,
                                 const Tensor& bn_b, const Tensor& bn_rm, const Tensor& bn_rv, std::vector<int64_t> tcfg,
                                 std::vector<double> tf, const c10::optional<Tensor>& g_nbt,
                                 const c10::optional<Tensor>& bn_nbt) {
        TORCH_CHECK(x_in.is_cuda() && x_in.dim() == 4, "cnsn_forward_bnrelu: expected an (N, C, H, W) HIP device tensor");
        const Config c = parse_config(cfg, fcfg);
        const bool want_y = tcfg[0] != 0, bn_training = tcfg[1] != 0;
        const c10::DeviceGuard device_guard(x_in.device());
        const Tensor x = dense(x_in);
        Tensor addend;
        if (c.add_mode != CNSN_ADD_NONE) {
            TORCH_CHECK(addend_in.has_value() && addend_in->is_cuda() && addend_in->sizes() == x.sizes() &&
                            addend_in->scalar_type() == x.scalar_type(),
                        "cnsn_forward_bnrelu: the addend must be a device tensor of x's shape and dtype");
            addend = dense(*addend_in);
        }
        cnsn_problem_t prob = make_problem(x, c);
        const cnsn_epilogue_t epi = make_epilogue(c, addend);
        const bool has_epi = c.add_mode != CNSN_ADD_NONE;
        const at::Device dev = x.device();
        GateTensors gg, bt;
        gg.init(g_w, g_gamma, g_beta, g_rm, g_rv, c.sn_training ? g_nbt : c10::nullopt);
        // (w slot unused: weight / bias / running buffers / counter of the BatchNorm2d)
        bt.init(bn_w, bn_w, bn_b, bn_rm, bn_rv, bn_training ? bn_nbt : c10::nullopt);
        cnsn_bn_tail_t tail{};
        tail.struct_bytes = (int32_t)sizeof(cnsn_bn_tail_t);
        tail.training = bn_training ? 1 : 0;
        tail.eps = (float)tf[0];
        tail.momentum = (float)tf[1];
        tail.weight = bt.gamma.data_ptr<float>();
        tail.bias = bt.beta.data_ptr<float>();
        tail.running_mean = bt.rm.data_ptr<float>();
        tail.running_var = bt.rv.data_ptr<float>();
        tail.num_batches_tracked = bt.c.num_batches_tracked;
        Tensor y = want_y ? out_like(x) : Tensor();
        Tensor z = out_like(x);
        const bool need_bwd = cfg[13] != 0;
        const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
        const size_t ws_bytes = cnsn_workspace_bytes(&prob);
        Tensor saved;
        if (need_bwd && ws_bytes > 0) saved = at::empty({(int64_t)cnsn_saved_floats(&prob)}, fopt);
        Tensor stats = at::empty({2 * x.size(1)}, fopt);
        Tensor ws = at::empty({(int64_t)(ws_bytes / 4) + 1}, fopt);
        hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        const int st = cnsn_forward_bnrelu(&prob, has_epi ? &epi : nullptr, &tail, x.data_ptr(), &gg.c,
                                           want_y ? y.data_ptr() : nullptr, z.data_ptr(),
                                           saved.defined() ? saved.data_ptr<float>() : nullptr, stats.data_ptr<float>(),
                                           ws.data_ptr(), ws_bytes, (void*)stream);
        check_status(st, "cnsn_forward_bnrelu");
        if (c.sn_training) gg.write_back();
        if (bn_training) bt.write_back();
        ctx->set_materialize_grads(false);  // an unused y hands an undefined tensor to the backward, not zeros
        if (need_bwd) {
            ctx->saved_data["cfg"] = cfg;
            ctx->saved_data["fcfg"] = fcfg;
            ctx->saved_data["tcfg"] = tcfg;
            ctx->saved_data["tf"] = tf;
            ctx->saved_data["pd"] = std::vector<int64_t>{(int64_t)g_w.scalar_type(), (int64_t)g_gamma.scalar_type(),
                                                         (int64_t)g_beta.scalar_type(), (int64_t)bn_w.scalar_type(),
                                                         (int64_t)bn_b.scalar_type()};
            Tensor none;
            ctx->save_for_backward({x, saved, addend.defined() ? addend : none, stats, gg.w, gg.gamma, gg.beta, gg.rm, gg.rv,
                                    bt.gamma, bt.beta, bt.rm, bt.rv});
        }
        return want_y ? variable_list{y, z} : variable_list{z};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto sv = ctx->get_saved_variables();
        const Tensor &x = sv[0], &saved = sv[1], &addend = sv[2], &stats = sv[3];
        const auto cfg = ctx->saved_data["cfg"].toIntVector();
        const auto fcfg = ctx->saved_data["fcfg"].toDoubleVector();
        const auto tcfg = ctx->saved_data["tcfg"].toIntVector();
        const auto tf = ctx->saved_data["tf"].toDoubleVector();
        const auto pd = ctx->saved_data["pd"].toIntVector();
        const Config c = parse_config(cfg, fcfg);
        const bool want_y = tcfg[0] != 0;
        cnsn_problem_t prob = make_problem(x, c);
        const at::Device dev = x.device();
        const c10::DeviceGuard device_guard(dev);
        Tensor gy = want_y ? grads[0] : Tensor();
        Tensor gz = want_y ? grads[1] : grads[0];
        if (!gz.defined()) gz = at::zeros_like(x);
        if (gz.scalar_type() != x.scalar_type()) gz = gz.to(x.scalar_type());
        gz = dense(gz);
        if (gy.defined()) {
            if (gy.scalar_type() != x.scalar_type()) gy = gy.to(x.scalar_type());
            gy = dense(gy);
        }
        cnsn_gate_t gg{};
        gg.fc_weight = sv[4].data_ptr<float>();
        gg.bn_weight = sv[5].data_ptr<float>();
        gg.bn_bias = sv[6].data_ptr<float>();
        gg.running_mean = sv[7].data_ptr<float>();
        gg.running_var = sv[8].data_ptr<float>();
        gg.num_batches_tracked = nullptr;
        cnsn_bn_tail_t tail{};
        tail.struct_bytes = (int32_t)sizeof(cnsn_bn_tail_t);
        tail.training = tcfg[1] != 0 ? 1 : 0;
        tail.eps = (float)tf[0];
        tail.momentum = (float)tf[1];
        tail.weight = sv[9].data_ptr<float>();
        tail.bias = sv[10].data_ptr<float>();
        tail.running_mean = sv[11].data_ptr<float>();
        tail.running_var = sv[12].data_ptr<float>();
        tail.num_batches_tracked = nullptr;
        const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
        const int64_t Cn = x.size(1);
        Tensor dx = out_like(x);
        Tensor flat = at::empty({6 * Cn}, fopt);  // dw (C,1,2) | dgamma | dbeta | d bn weight | d bn bias
        cnsn_gate_grad_t dg{flat.data_ptr<float>(), flat.data_ptr<float>() + 2 * Cn, flat.data_ptr<float>() + 3 * Cn};
        const size_t ws_bytes = cnsn_workspace_bytes(&prob);
        Tensor ws = at::empty({(int64_t)(ws_bytes / 4) + 1}, fopt);
        const cnsn_epilogue_t epi = make_epilogue(c, addend);
        hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        const int st = cnsn_backward_bnrelu(&prob, c.add_mode != CNSN_ADD_NONE ? &epi : nullptr, &tail,
                                            gy.defined() ? gy.data_ptr() : nullptr, gz.data_ptr(), x.data_ptr(), &gg,
                                            saved.data_ptr<float>(), stats.data_ptr<float>(), dx.data_ptr(), &dg,
                                            flat.data_ptr<float>() + 4 * Cn, flat.data_ptr<float>() + 5 * Cn, ws.data_ptr(),
                                            ws_bytes, (void*)stream);
        check_status(st, "cnsn_backward_bnrelu");
        auto cast = [](const Tensor& t, int64_t code) { return (int64_t)t.scalar_type() != code ? t.to((at::ScalarType)code) : t; };
        Tensor none;
        //               x   cfg   fcfg  g_w g_gamma g_beta g_rm g_rv addend bn_w bn_b bn_rm bn_rv tcfg tf g_nbt bn_nbt
        variable_list out(17, none);
        out[0] = dx;
        out[3] = cast(flat.narrow(0, 0, 2 * Cn).view({Cn, 1, 2}), pd[0]);
        out[4] = cast(flat.narrow(0, 2 * Cn, Cn), pd[1]);
        out[5] = cast(flat.narrow(0, 3 * Cn, Cn), pd[2]);
        if (c.add_mode == CNSN_ADD_PRE) out[8] = dx;
        out[9] = cast(flat.narrow(0, 4 * Cn, Cn), pd[3]);
        out[10] = cast(flat.narrow(0, 5 * Cn, Cn), pd[4]);
        return out;
    }
};

std::vector<Tensor> fused_cnsn_tail(const Tensor& x, std::vector<int64_t> cfg, std::vector<double> fcfg, const Tensor& g_w,
                                    const Tensor& g_gamma, const Tensor& g_beta, const Tensor& g_rm, const Tensor& g_rv,
                                    const c10::optional<Tensor>& addend, const Tensor& bn_w, const Tensor& bn_b,
                                    const Tensor& bn_rm, const Tensor& bn_rv, std::vector<int64_t> tcfg,
                                    std::vector<double> tf, const c10::optional<Tensor>& g_nbt,
                                    const c10::optional<Tensor>& bn_nbt) {
    return FusedCNSNTail::apply(x, cfg, fcfg, g_w, g_gamma, g_beta, g_rm, g_rv, addend, bn_w, bn_b, bn_rm, bn_rv, tcfg, tf, g_nbt,
                                bn_nbt);
}

int64_t bnrelu_plan(const Tensor& x, std::vector<int64_t> cfg, std::vector<double> fcfg, bool backward) {
    const Config c = parse_config(cfg, fcfg);
    cnsn_problem_t prob = make_problem(x, c);
    Tensor none;
    const cnsn_epilogue_t epi = make_epilogue(c, none);
    return cnsn_bnrelu_plan(&prob, c.add_mode != CNSN_ADD_NONE ? &epi : nullptr, backward ? 1 : 0);
}

Tensor fused_cnsn(const Tensor& x, std::vector<int64_t> cfg, std::vector<double> fcfg, const c10::optional<Tensor>& perm,
                  const c10::optional<Tensor>& chan, const c10::optional<Tensor>& g_w,
                  const c10::optional<Tensor>& g_gamma, const c10::optional<Tensor>& g_beta,
                  const c10::optional<Tensor>& g_rm, const c10::optional<Tensor>& g_rv, const c10::optional<Tensor>& f_w,
                  const c10::optional<Tensor>& f_gamma, const c10::optional<Tensor>& f_beta,
                  const c10::optional<Tensor>& f_rm, const c10::optional<Tensor>& f_rv,
                  const c10::optional<Tensor>& addend, const c10::optional<Tensor>& g_nbt,
                  const c10::optional<Tensor>& f_nbt) {
    return FusedCNSN::apply(x, cfg, fcfg, perm, chan, g_w, g_gamma, g_beta, g_rm, g_rv, f_w, f_gamma, f_beta, f_rm, f_rv,
                            addend, g_nbt, f_nbt);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "C++ autograd glue over the C ABI of libcnsn_hip.so";
    m.def("fused_cnsn", &fused_cnsn, "fused CrossNorm+SelfNorm forward (autograd-aware)");
    m.def("fused_cnsn_tail", &fused_cnsn_tail, "SelfNorm + the next block's BatchNorm2d + ReLU in one launch (autograd-aware)");
    m.def("bnrelu_plan", &bnrelu_plan, "1 when a fused kernel takes the call");
    m.def("abi_version", []() { return cnsn_abi_version(); });
    m.def("arena_config", [](int64_t min_bytes) {
        const int64_t was = arena_min_bytes();
        g_arena_min.store(min_bytes < 0 ? -1 : min_bytes, std::memory_order_relaxed);
        return was;
    }, "threshold in bytes from which the op's outputs come from the output arena (< 0: never); returns the previous value");
    m.def("arena_min_bytes", []() { return arena_min_bytes(); });
    m.def("arena_empty_like", [](const Tensor& x) {
        TORCH_CHECK(x.is_cuda(), "arena_empty_like: device tensors only");
        Tensor t = arena_empty(x.sizes(), at::detail::defaultStrides(x.sizes()), at::TensorOptions().dtype(x.scalar_type()).device(x.device()),
                               x.device(), x.numel() * (int64_t)x.element_size());
        return t.defined() ? t : at::empty(x.sizes(), x.options().memory_format(at::MemoryFormat::Contiguous));
    }, "a fresh contiguous tensor of x's shape and type over an arena block (torch's allocator when the arena cannot serve it)");
    m.def("arena_trim", [](int64_t device) { arena_pool_trim((int)device); },
          "release the free cached blocks of the arena's torch pool (device < 0: every device)");
    m.def("arena_pool_id", [](int64_t device) {
        std::shared_ptr<at::cuda::MemPool> p = arena_pool((int)device);
        return p ? std::make_pair((int64_t)p->id().first, (int64_t)p->id().second) : std::make_pair((int64_t)0, (int64_t)0);
    }, "id of the torch MemPool behind the arena on `device` ((0, 0): none) — for torch.cuda.memory_snapshot(id)");
    m.def("out_like", &out_like, "the op's output allocation for a dense x: arena from the threshold on, else torch");
}
