// Output arena: where the op's y / dx (and the fused block's z) live.
//
// The reference's op returns NEW tensors (models/cnsn.py:29,150: `instance_norm_mix(...)`, `x * g`), allocated by torch's
// caching allocator.  On MI355X the speed of the single-touch launches' plane-strided WRITES depends on the physical placement
// of their target (profiles/r04_memory_map.md): one large hipMalloc'ed block in five lies where writes run at the copy rate,
// the rest 10-20 % below it, and nothing user space can say to the allocator changes where a block lands.  Address ranges
// MAPPED from many small physical allocations can be created several at a time, TIMED and the fastest kept
// (profiles/r05_arena.md): this file hands out such ranges — hipMemCreate chunks of `chunk_bytes` (default 56 MiB;
// CNSN_ARENA_CHUNK_MB) plus one tail chunk of the remainder, hipMemAddressReserve + hipMemMap + hipMemSetAccess.
//
// Two ways in (include/cnsn_hip.h, "output arena"):
//  * cnsn_arena_map / cnsn_arena_unmap (ABI 7) — create / release ONE block, nothing cached here.  They have the signature of
//    a torch pluggable allocator: the Python layer's arena is a `torch.cuda.MemPool` over these two functions
//    (csrc/glue/cnsn_glue.cpp), so the blocks ARE the caching allocator's: it caches and splits them, `memory_allocated`
//    counts them, `Tensor.record_stream` works on them, its out-of-memory path releases them (and, `use_on_oom`, lends the
//    pool's free blocks to any other allocation that would otherwise fail).  Round 5 handed out `at::from_blob` tensors over
//    a cache of this file's own, which torch could neither see, trim nor order across streams (review of round 5).
//  * cnsn_arena_alloc / cnsn_arena_free — the same blocks behind a small caching layer of the library's own, for callers of
//    the C ABI that have no allocator to plug into: per-size free lists, a CAP on what the arena holds (cnsn_arena_set_limit,
//    default half of the device memory) enforced by evicting the least recently used free blocks, best fit within 1/8 of the
//    request, and trim-and-retry when the driver has no memory left.
//
// Stream semantics of the caching layer = a caching allocator's: a block freed after work was queued on stream S may be
// handed out again at once to a request on S (stream order protects it); a request on ANOTHER stream first waits for
// everything queued on S so far (event recorded at that moment) and for every stream named by cnsn_arena_record_stream.
// Nothing here is used while a stream is being captured into a graph: a replay must find its tensors at fixed addresses.
//
// Two findings of round 5 shape the code (profiles/r05_arena.md):
//  * WHERE a block lies physically decides how fast it is written, and nothing else does — not the chunk size it is
//    composed of (2 MiB ... one chunk per block: about one block in five is fast with every size).  So a block can be TIMED
//    when it is created (`arena_write_probe`: a plane-strided fill, ~1 ms); `cnsn_arena_prospect` creates more candidates than
//    it keeps — the bounded, explicit form of "look for fast memory".
//  * An address range that was unmapped must NEVER be mapped again: on ROCm 7.2 a range re-reserved after
//    hipMemUnmap + hipMemAddressFree and mapped to other physical memory was read and written through stale translations
//    (y and dx of a launch landed in each other's old pages: tests/test_gpu_arena.py, the trim test).  Released blocks
//    therefore give back their PHYSICAL memory only; their address range stays reserved for the life of the process
//    (addresses are plentiful: 2^47).
//
// Not part of the numerical path: the launches write the same bytes wherever their output lies
// (tests/test_gpu_arena.py::test_results_do_not_depend_on_the_arena).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/cnsn_hip.h"
#include "cnsn_env.h"

namespace cnsn {
namespace {

constexpr size_t kMiB = size_t(1) << 20;
constexpr size_t kDefaultChunk = 56 * kMiB;
constexpr int kDefaultTries = 8;
// Blocks below this size are never timed: a write of up to ~256 MB is absorbed by the Infinity Cache and says nothing about the
// memory behind it (392 single 56 MiB chunks all 5.9 TB/s, 112 blocks of 224 MiB all 7.8 TB/s, 784 MiB blocks 5.3 or 6.8:
// profiles/r05_arena.md section 5)
constexpr size_t kTimedFrom = 384 * kMiB;

struct Piece {
    hipMemGenericAllocationHandle_t handle;
    size_t bytes;
};

struct Block {
    void* va = nullptr;
    size_t bytes = 0;     // mapped size: whole chunks + one tail chunk (a multiple of the granularity)
    size_t reserved = 0;  // size of the address range (= bytes once mapped)
    size_t chunk = 0;     // chunk size in force when it was created
    int device = 0;
    hipStream_t stream = nullptr;          // the stream of the request it was last handed out to
    std::vector<hipStream_t> also;         // streams named by cnsn_arena_record_stream since then
    std::vector<Piece> pieces;
    bool in_use = false;
    bool pooled = false;  // created by cnsn_arena_map: cached by the caller (torch's allocator), never on a free list here
    float gbps = 0.f;     // measured write rate (arena_write_probe), 0 = never measured
    uint64_t tick = 0;    // when it was last freed (least recently used goes first when the cap bites)
};

struct DeviceArena {
    std::multimap<size_t, Block*> free_by_size;
    uint64_t mapped = 0, in_use = 0, blocks = 0, blocks_in_use = 0, hits = 0, misses = 0, failed = 0, probed = 0, evicted = 0;
    uint64_t limit = 0;       // cap on `mapped` of the caching layer; 0: not resolved yet
    bool limit_set = false;   // by cnsn_arena_set_limit (not the default)
    bool broken = false;      // the driver refused the virtual-memory calls on this device (not: ran out of memory)
};

struct Arena {
    std::mutex mu;
    std::unordered_map<void*, Block*> by_ptr;
    std::map<int, DeviceArena> dev;
    size_t chunk = 0;       // resolved on first use
    size_t granularity = 0;
    int tries = -1;         // candidates per new block (cnsn_arena_set_tries; -1: CNSN_ARENA_TRIES, default 8)
    uint64_t tick = 0;
};

Arena& arena() {
    static Arena* a = new Arena;  // (leaked on purpose: tensors may be released after static destructors have run)
    return *a;
}

size_t round_up(size_t v, size_t to) { return (v + to - 1) / to * to; }

struct DeviceScope {  // make `device` current for the calls that need it, put the caller's back afterwards
    int was = -1, dev;
    explicit DeviceScope(int device) : dev(device) {
        (void)hipGetDevice(&was);
        if (was != dev) (void)hipSetDevice(dev);
    }
    ~DeviceScope() {
        if (was != dev && was >= 0) (void)hipSetDevice(was);
    }
};

int resolve_tries() {  // (under the arena's mutex)
    Arena& a = arena();
    if (a.tries < 0) {
        const char* e = knob(K_ARENA_TRIES);
        a.tries = (e && atoi(e) > 0) ? atoi(e) : kDefaultTries;
        if (a.tries > 32) a.tries = 32;
    }
    return a.tries;
}

size_t resolve_chunk(Arena& a, int device) {
    if (a.chunk) return a.chunk;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) {
        (void)hipGetLastError();
        gran = 2 * kMiB;
    }
    a.granularity = gran;
    size_t want = kDefaultChunk;
    if (const char* e = knob(K_ARENA_CHUNK_MB))
        if (atoll(e) > 0) want = (size_t)atoll(e) * kMiB;
    a.chunk = round_up(want, gran);
    return a.chunk;
}

// what the caching layer may hold on `device`: cnsn_arena_set_limit, else CNSN_ARENA_MAX_MB, else half of the device memory
uint64_t resolve_limit(DeviceArena& d) {  // (device current)
    if (d.limit) return d.limit;
    if (const char* e = knob(K_ARENA_MAX_MB))
        if (atoll(e) > 0) return d.limit = (uint64_t)atoll(e) * kMiB;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) {
        (void)hipGetLastError();
        return ~uint64_t(0);  // (unknown: no cap this time, ask again)
    }
    return d.limit = total_b / 2;
}

// physical memory back to the driver; the address range stays reserved (see the header: never map a range twice)
void release_block(Block* b) {
    if (b->va && b->bytes) {
        size_t off = 0;
        for (const Piece& p : b->pieces) {  // mapping by mapping
            (void)hipMemUnmap((char*)b->va + off, p.bytes);
            off += p.bytes;
        }
    }
    for (const Piece& p : b->pieces) (void)hipMemRelease(p.handle);
    (void)hipGetLastError();
    delete b;
}

// ---- how fast is a block written?  A fill in the cluster kernels' order of accesses: waves write runs of 12 KiB (a 56x56
// fp32 plane is 12.25 KiB), and the runs in flight at one time lie bytes/256 apart (the planes of one channel over a batch of
// 256).  Relative numbers are all that matters: the ranking agrees with the library's own launches (tools/arena_probe.py).
constexpr int kProbeRun = 12 * 1024;
__global__ __launch_bounds__(256) void arena_write_probe(char* base, size_t runs) {
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t cols = runs >= 256 ? runs / 256 : 1, full = cols * (runs >= 256 ? 256 : runs);
    const uint4 zero = {0u, 0u, 0u, 0u};
    for (size_t r = w; r < runs; r += waves) {
        const size_t loc = r < full ? (r % (full / cols)) * cols + r / (full / cols) : r;
        uint4* p = (uint4*)(base + loc * kProbeRun) + lane;
#pragma unroll
        for (int j = 0; j < kProbeRun / 1024; ++j) p[j * 64] = zero;
    }
}

// GB/s of the fill on `stream` (synchronises it); 0 when the measurement could not be made
float measure_block(const Block* b, hipStream_t stream) {
    const size_t runs = b->bytes / kProbeRun;
    if (runs == 0) return 0.f;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        (void)hipGetLastError();
        return 0.f;
    }
    const int grid = 2048, reps = 3;
    arena_write_probe<<<grid, 256, 0, stream>>>((char*)b->va, runs);  // (first touch: page-table walks, not the rate)
    (void)hipEventRecord(e0, stream);
    for (int i = 0; i < reps; ++i) arena_write_probe<<<grid, 256, 0, stream>>>((char*)b->va, runs);
    (void)hipEventRecord(e1, stream);
    float ms = 0.f;
    const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok) {
        (void)hipGetLastError();
        return 0.f;
    }
    return (float)((double)runs * kProbeRun * reps / ((double)ms * 1e6));
}

// A new block of `bytes` (a multiple of the granularity) on `device` (current), or nullptr.  `*err` tells the two reasons apart:
// hipErrorOutOfMemory — the device is full, the caller may make room and ask again — against anything else, which means the
// virtual-memory calls do not work here.
Block* create_block(Arena& a, int device, size_t bytes, hipError_t* err) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    Block* b = new Block;
    b->device = device;
    b->chunk = a.chunk;
    *err = hipSuccess;
    for (size_t off = 0; off < bytes;) {  // whole chunks, then one tail chunk: a 57 MiB request holds 58 MiB, not 112
        const size_t piece = std::min(a.chunk, bytes - off);
        hipMemGenericAllocationHandle_t h;
        const hipError_t e = hipMemCreate(&h, piece, &prop, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            *err = e;
            release_block(b);
            return nullptr;
        }
        b->pieces.push_back({h, piece});
        off += piece;
    }
    hipError_t e = hipMemAddressReserve(&b->va, bytes, 0, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *err = e;
        b->va = nullptr;
        release_block(b);
        return nullptr;
    }
    b->reserved = bytes;
    size_t off = 0;
    for (size_t i = 0; i < b->pieces.size(); ++i) {
        e = hipMemMap((char*)b->va + off, b->pieces[i].bytes, 0, b->pieces[i].handle, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            *err = e;
            size_t o2 = 0;
            for (size_t q = 0; q < i; ++q) {
                (void)hipMemUnmap((char*)b->va + o2, b->pieces[q].bytes);
                o2 += b->pieces[q].bytes;
            }
            release_block(b);  // (bytes is still 0: nothing more to unmap)
            return nullptr;
        }
        off += b->pieces[i].bytes;
    }
    b->bytes = bytes;  // (from here on release_block unmaps the whole range)
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(b->va, bytes, &acc, 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *err = e;
        release_block(b);
        return nullptr;
    }
    return b;
}

// The arena's standing policy for a NEW block (device current, arena locked).  Best of `tries`: WHERE a block lies physically
// decides how fast it is written (about one candidate in five is of the fast kind), so a new block of kTimedFrom bytes or more
// is chosen among a few candidates created together — distinct physical memory —, each timed with the plane-strided fill
// (~1 ms), the losers' memory handed back at once.  Happens when a block is CREATED (the first steps of a job), costs `tries` x
// the block's size transiently, never more than a quarter of what is free.  Nothing is timed on a capturing stream.
Block* new_block(Arena& a, DeviceArena& d, int device, size_t need, hipStream_t stream, hipError_t* err) {
    *err = hipSuccess;
    int tries = need >= kTimedFrom ? resolve_tries() : 1;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            if (need > free_b) {  // (no point in creating thousands of chunks to find that out)
                *err = hipErrorOutOfMemory;
                return nullptr;
            }
            if (tries > 1) tries = (int)std::max<size_t>(1, std::min<size_t>((size_t)tries, free_b / 4 / need));
        } else {
            (void)hipGetLastError();
        }
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (tries > 1 && (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)) {
        (void)hipGetLastError();
        tries = 1;
    }
    // CNSN_ARENA_SPREAD_GB (A/B knob, default 0): candidates created one after the other are neighbours in physical memory,
    // and the fast regions are tens of GB wide — with this many GB of physical allocations held BETWEEN the candidates (never
    // mapped, given back with the losers) they sample different stretches of the device memory
    std::vector<Block*> cand;
    std::vector<hipMemGenericAllocationHandle_t> spacers;
    size_t spacer_each = 0;
    if (tries > 1) {
        if (const char* e = knob(K_ARENA_SPREAD_GB))
            if (atof(e) > 0) spacer_each = (size_t)(atof(e) * 1073741824.0) / (size_t)(tries - 1);
    }
    hipMemAllocationProp sprop{};
    sprop.type = hipMemAllocationTypePinned;
    sprop.location.type = hipMemLocationTypeDevice;
    sprop.location.id = device;
    for (int i = 0; i < tries; ++i) {
        if (i > 0 && spacer_each) {
            const size_t piece = round_up(size_t(1) << 30, a.granularity ? a.granularity : 2 * kMiB);
            for (size_t got = 0; got < spacer_each; got += piece) {
                hipMemGenericAllocationHandle_t h;
                if (hipMemCreate(&h, piece, &sprop, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    break;
                }
                spacers.push_back(h);
            }
        }
        hipError_t e = hipSuccess;
        Block* c = create_block(a, device, need, &e);
        if (!c) {
            if (cand.empty()) *err = e;
            break;
        }
        cand.push_back(c);
    }
    if (cand.size() > 1) {
        for (Block* c : cand) c->gbps = measure_block(c, stream);
        std::sort(cand.begin(), cand.end(), [](const Block* x, const Block* y) { return x->gbps > y->gbps; });
        // (keeping a fast runner-up for the next request of the size was tried: the second-best of eight is slower than the best
        // of the next eight — 0.766-0.772 against 0.759 ms per headline step on one box; not kept)
        for (size_t i = 1; i < cand.size(); ++i) release_block(cand[i]);  // (measure_block left the stream idle)
        d.probed += cand.size();
    }
    for (auto h : spacers) (void)hipMemRelease(h);
    return cand.empty() ? nullptr : cand[0];
}

// everything queued so far on the streams that used `b` comes before what `stream` does next (device current)
void order_behind_previous_users(Block* b, hipStream_t stream) {
    auto wait_for = [&](hipStream_t s) {
        if (s == stream) return;
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
            const bool ok = hipEventRecord(ev, s) == hipSuccess && hipStreamWaitEvent(stream, ev, 0) == hipSuccess;
            (void)hipEventDestroy(ev);  // (released once the wait has been satisfied)
            if (ok) return;
        }
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();  // (a destroyed stream, no events left: the safe form of the same order)
        (void)hipGetLastError();
    };
    wait_for(b->stream);
    for (hipStream_t s : b->also) wait_for(s);
    b->also.clear();
}

// take blocks off the free list of `d`, least recently freed first, until `room` more bytes fit under `cap` (or none is left);
// the caller settles their streams and releases them
void evict_lru(Arena& a, DeviceArena& d, uint64_t room, uint64_t cap, std::vector<Block*>* out) {
    while (d.mapped + room > cap && !d.free_by_size.empty()) {
        auto oldest = d.free_by_size.begin();
        for (auto it = d.free_by_size.begin(); it != d.free_by_size.end(); ++it)
            if (it->second->tick < oldest->second->tick) oldest = it;
        Block* b = oldest->second;
        d.free_by_size.erase(oldest);
        d.mapped -= b->bytes;
        --d.blocks;
        ++d.evicted;
        a.by_ptr.erase(b->va);
        out->push_back(b);
    }
}

void settle_and_release(std::vector<Block*>& drop) {  // (device current, may run under the arena's mutex: the rare path)
    for (Block* b : drop) {  // work queued on the block's streams may still touch it
        bool ok = hipStreamSynchronize(b->stream) == hipSuccess;
        for (hipStream_t s : b->also) ok = (hipStreamSynchronize(s) == hipSuccess) && ok;
        if (!ok) {
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
        }
        release_block(b);
    }
    drop.clear();
}

void note_failure(DeviceArena& d, hipError_t e) {
    ++d.failed;
    // "does not work here" is remembered per device; "no memory now" is not (round 5 turned the arena off for the whole
    // process after two failed requests of any kind on one device)
    if (e != hipSuccess && e != hipErrorOutOfMemory && d.blocks == 0) d.broken = true;
}

}  // namespace
}  // namespace cnsn

using namespace cnsn;

extern "C" {

void* cnsn_arena_alloc(int device, size_t bytes, void* stream_) {
    if (bytes == 0 || device < 0) return nullptr;
    hipStream_t stream = (hipStream_t)stream_;
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    DeviceArena& d = a.dev[device];
    if (d.broken) return nullptr;
    DeviceScope on(device);
    const size_t chunk = resolve_chunk(a, device);
    const size_t need = round_up(bytes, a.granularity);
    Block* b = nullptr;
    // A free block of this size — the fastest measured one, among equals one last used on the requesting stream — or, failing
    // that, the smallest free block that wastes no more than an eighth of the request.
    auto pick = d.free_by_size.end();
    for (auto it = d.free_by_size.lower_bound(need); it != d.free_by_size.end() && it->first <= need + need / 8; ++it) {
        if (it->second->chunk != chunk) continue;  // (a block from before cnsn_arena_set_chunk_bytes)
        if (pick == d.free_by_size.end()) {
            pick = it;
            continue;
        }
        if (it->first != pick->first) break;  // (sizes ascend: the first size that has a block wins)
        const Block *have = pick->second, *cand = it->second;
        if (cand->gbps > have->gbps || (cand->gbps == have->gbps && cand->stream == stream && have->stream != stream)) pick = it;
    }
    if (pick != d.free_by_size.end()) {
        b = pick->second;
        d.free_by_size.erase(pick);
        ++d.hits;
        order_behind_previous_users(b, stream);
    } else {
        std::vector<Block*> drop;
        const uint64_t cap = resolve_limit(d);
        if (d.mapped + need > cap) {  // the cap: least recently used free blocks make room
            uint64_t idle = 0;
            for (auto& kv : d.free_by_size) idle += kv.first;
            if (d.mapped - idle + need > cap) {  // (not even with every free block gone: the caller allocates as it always did)
                ++d.failed;
                return nullptr;
            }
            evict_lru(a, d, need, cap, &drop);
            settle_and_release(drop);
        }
        hipError_t err = hipSuccess;
        b = new_block(a, d, device, need, stream, &err);
        if (!b && err == hipErrorOutOfMemory && !d.free_by_size.empty()) {  // the device is full: give back what lies idle, once
            uint64_t idle = 0;
            for (auto& kv : d.free_by_size) idle += kv.first;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) (void)hipGetLastError();
            if (need <= free_b + idle) {  // (... unless the request could not be met even then)
                evict_lru(a, d, ~uint64_t(0) / 2, 0, &drop);
                settle_and_release(drop);
                b = new_block(a, d, device, need, stream, &err);
            }
        }
        if (!b) {
            note_failure(d, err);
            return nullptr;
        }
        ++d.misses;
        ++d.blocks;
        d.mapped += b->bytes;
        a.by_ptr[b->va] = b;
    }
    b->stream = stream;
    b->in_use = true;
    d.in_use += b->bytes;
    ++d.blocks_in_use;
    return b->va;
}

int cnsn_arena_free(void* ptr) {
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    auto it = a.by_ptr.find(ptr);
    if (it == a.by_ptr.end() || !it->second->in_use || it->second->pooled) return CNSN_E_NULL;
    Block* b = it->second;
    DeviceArena& d = a.dev[b->device];
    b->in_use = false;
    b->tick = ++a.tick;
    d.in_use -= b->bytes;
    --d.blocks_in_use;
    d.free_by_size.emplace(b->bytes, b);
    return CNSN_OK;
}

int cnsn_arena_record_stream(void* ptr, void* stream) {
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    auto it = a.by_ptr.find(ptr);
    if (it == a.by_ptr.end() || !it->second->in_use || it->second->pooled) return CNSN_E_NULL;
    Block* b = it->second;
    if ((hipStream_t)stream != b->stream && std::find(b->also.begin(), b->also.end(), (hipStream_t)stream) == b->also.end())
        b->also.push_back((hipStream_t)stream);
    return CNSN_OK;
}

void* cnsn_arena_map(size_t bytes, int device, void* stream) {
    if (bytes == 0 || device < 0) return nullptr;
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    DeviceArena& d = a.dev[device];
    if (d.broken) return nullptr;
    DeviceScope on(device);
    (void)resolve_chunk(a, device);
    const size_t need = round_up(bytes, a.granularity);
    hipError_t err = hipSuccess;
    Block* b = new_block(a, d, device, need, (hipStream_t)stream, &err);
    if (!b) {  // (the caller — torch's allocator — frees its caches and asks again, or reports the out-of-memory condition)
        note_failure(d, err);
        return nullptr;
    }
    b->pooled = true;
    b->in_use = true;
    b->stream = (hipStream_t)stream;
    ++d.misses;
    ++d.blocks;
    ++d.blocks_in_use;
    d.mapped += b->bytes;
    d.in_use += b->bytes;
    a.by_ptr[b->va] = b;
    return b->va;
}

void cnsn_arena_unmap(void* ptr, size_t /*bytes*/, int /*device*/, void* /*stream*/) {
    Arena& a = arena();
    Block* b = nullptr;
    {
        std::lock_guard<std::mutex> lock(a.mu);
        auto it = a.by_ptr.find(ptr);
        if (it == a.by_ptr.end() || !it->second->pooled) return;
        b = it->second;
        DeviceArena& d = a.dev[b->device];
        d.mapped -= b->bytes;
        d.in_use -= b->bytes;
        --d.blocks;
        --d.blocks_in_use;
        a.by_ptr.erase(it);
    }
    DeviceScope on(b->device);
    release_block(b);  // (the owner of the cache has already ordered this behind every use of the block)
}

int cnsn_arena_owns(const void* ptr) {
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    for (auto& kv : a.by_ptr) {
        const char* lo = (const char*)kv.first;
        if ((const char*)ptr >= lo && (const char*)ptr < lo + kv.second->bytes) return 1;
    }
    return 0;
}

size_t cnsn_arena_trim(int device) {
    Arena& a = arena();
    std::vector<Block*> drop;
    {
        std::lock_guard<std::mutex> lock(a.mu);
        for (auto& dv : a.dev) {
            if (device >= 0 && dv.first != device) continue;
            for (auto& kv : dv.second.free_by_size) drop.push_back(kv.second);
            dv.second.free_by_size.clear();
        }
        for (Block* b : drop) {
            DeviceArena& d = a.dev[b->device];
            d.mapped -= b->bytes;
            --d.blocks;
            a.by_ptr.erase(b->va);
        }
    }
    size_t freed = 0;
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (Block* b : drop) {
        (void)hipSetDevice(b->device);
        freed += b->bytes;
        std::vector<Block*> one{b};
        settle_and_release(one);
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    return freed;
}

int cnsn_arena_set_limit(int device, uint64_t bytes) {
    if (device < 0) return CNSN_E_NULL;
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    DeviceArena& d = a.dev[device];
    d.limit = bytes;  // (0: back to CNSN_ARENA_MAX_MB / half of the device memory, resolved at the next request)
    d.limit_set = bytes != 0;
    if (bytes) {
        DeviceScope on(device);
        std::vector<Block*> drop;
        evict_lru(a, d, 0, bytes, &drop);
        settle_and_release(drop);
    }
    return CNSN_OK;
}

int cnsn_arena_prospect(int device, size_t bytes, int keep, int candidates, void* stream, float* gbps_out) {
    if (bytes == 0 || device < 0 || keep < 0 || candidates <= 0) return CNSN_E_NULL;
    Arena& a = arena();
    DeviceScope on(device);
    std::vector<Block*> made;
    size_t need = 0;
    {
        std::lock_guard<std::mutex> lock(a.mu);
        if (a.dev[device].broken) return 0;
        (void)resolve_chunk(a, device);
        need = round_up(bytes, a.granularity);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)  // never more than half of what is free
            candidates = (int)std::min<size_t>((size_t)candidates, free_b / 2 / need);
        else
            (void)hipGetLastError();
        for (int i = 0; i < candidates; ++i) {  // all alive at once: distinct physical memory
            hipError_t e = hipSuccess;
            Block* b = create_block(a, device, need, &e);
            if (!b) break;
            made.push_back(b);
        }
    }
    for (Block* b : made) b->gbps = measure_block(b, (hipStream_t)stream);
    if (gbps_out)
        for (int i = 0; i < candidates; ++i) gbps_out[i] = i < (int)made.size() ? made[i]->gbps : 0.f;
    std::vector<Block*> order(made);
    std::sort(order.begin(), order.end(), [](const Block* x, const Block* y) { return x->gbps > y->gbps; });
    int kept = 0;
    {
        std::lock_guard<std::mutex> lock(a.mu);
        DeviceArena& d = a.dev[device];
        for (Block* b : order) {
            if (kept >= keep) break;
            b->stream = (hipStream_t)stream;
            b->tick = ++a.tick;
            ++d.blocks;
            d.mapped += b->bytes;
            a.by_ptr[b->va] = b;
            d.free_by_size.emplace(b->bytes, b);
            ++kept;
        }
    }
    (void)hipStreamSynchronize((hipStream_t)stream);
    for (size_t i = (size_t)kept; i < order.size(); ++i) release_block(order[i]);
    return kept;
}

int cnsn_arena_block_gbps(const void* ptr, float* gbps) {
    if (!gbps) return CNSN_E_NULL;
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    for (auto& kv : a.by_ptr) {  // (any address inside a block: torch's allocator may have split it)
        const char* lo = (const char*)kv.first;
        if ((const char*)ptr >= lo && (const char*)ptr < lo + kv.second->bytes) {
            *gbps = kv.second->gbps;
            return CNSN_OK;
        }
    }
    return CNSN_E_NULL;
}

int cnsn_arena_stats(int device, cnsn_arena_stats_t* out) {
    if (!out || out->struct_bytes != (int32_t)sizeof(cnsn_arena_stats_t)) return CNSN_E_NULL;
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    const DeviceArena& d = a.dev[device];
    out->device = device;
    out->chunk_bytes = a.chunk;
    out->mapped_bytes = d.mapped;
    out->in_use_bytes = d.in_use;
    out->blocks = d.blocks;
    out->blocks_in_use = d.blocks_in_use;
    out->hits = d.hits;
    out->misses = d.misses;
    out->failed = d.failed;
    out->probed = d.probed;
    out->tries = (uint64_t)(a.tries < 0 ? 0 : a.tries);
    out->evicted = d.evicted;
    out->limit_bytes = d.limit;
    out->broken = d.broken ? 1 : 0;
    return CNSN_OK;
}

int cnsn_arena_set_tries(int tries) {
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    const int was = resolve_tries();
    a.tries = tries < 1 ? -1 : (tries > 32 ? 32 : tries);
    return was;
}

int cnsn_arena_set_chunk_bytes(size_t chunk_bytes) {
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    a.chunk = 0;
    if (chunk_bytes) {
        if (!a.granularity) a.granularity = 2 * kMiB;
        a.chunk = round_up(chunk_bytes, a.granularity);
    }
    for (auto& dv : a.dev) dv.second.broken = false;
    return CNSN_OK;
}

}  // extern "C"
