// Output arena: where the op's y / dx (and the fused block's z) live.
//
// The reference's op returns NEW tensors (models/cnsn.py:29,150: `instance_norm_mix(...)`, `x * g`), allocated by torch's
// caching allocator.  On MI355X the speed of the single-touch launches' plane-strided WRITES depends on the physical placement
// of their target (profiles/r04_memory_map.md): one large hipMalloc'ed block in five lies where writes run at the copy rate,
// the rest 10-20 % below it, and nothing user space can say to the allocator changes where a block lands.  Address ranges
// MAPPED from many small physical allocations do not have that lottery (profiles/r05_arena.md): this arena hands out such
// ranges — hipMemCreate chunks of `chunk_bytes` (default 56 MiB; CNSN_ARENA_CHUNK_MB), hipMemAddressReserve + hipMemMap +
// hipMemSetAccess — and keeps freed blocks mapped on a per-size free list, so the steady state of a training loop is a
// mutex, a list pop and nothing else (no driver call, no synchronisation).
//
// Stream semantics = a caching allocator's: a block freed after work was queued on stream S may be handed out again at
// once to a request on S (stream order protects it); a request on ANOTHER stream first waits for everything queued on S so
// far (event recorded at that moment).  The callers (glue / functional.py) do not use the arena while their stream is
// being captured into a graph: a replay must find its tensors at fixed addresses that nobody else re-uses.
//
// Two findings of round 5 shape the code (profiles/r05_arena.md):
//  * WHERE a block lies physically decides how fast it is written, and nothing else does — not the chunk size it is
//    composed of (2 MiB ... one chunk per block: about one block in five is fast with every size).  So a block can be TIMED
//    when it is created (`arena_write_probe`: a plane-strided fill, ~1 ms) and the free lists hand out the fastest block of a
//    size first; `cnsn_arena_prospect` creates more candidates than it keeps — the bounded, explicit form of "look for fast
//    memory" (nothing does it by default).
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

struct Block {
    void* va = nullptr;
    size_t bytes = 0;  // mapped size (a whole number of chunks)
    size_t chunk = 0;  // size of the physical allocations it is mapped from
    int device = 0;
    hipStream_t stream = nullptr;  // the stream of the request it was last handed out to
    std::vector<hipMemGenericAllocationHandle_t> chunks;
    bool in_use = false;
    float gbps = 0.f;  // measured write rate (arena_write_probe), 0 = never measured
};

struct DeviceArena {
    std::multimap<size_t, Block*> free_by_size;
    uint64_t mapped = 0, in_use = 0, blocks = 0, blocks_in_use = 0, hits = 0, misses = 0, failed = 0, probed = 0;
};

struct Arena {
    std::mutex mu;
    std::unordered_map<void*, Block*> by_ptr;
    std::map<int, DeviceArena> dev;
    size_t chunk = 0;       // resolved on first use
    size_t granularity = 0;
    bool broken = false;    // the driver refused the virtual-memory calls once: never try again
    int tries = -1;         // candidates per new block (cnsn_arena_set_tries; -1: CNSN_ARENA_TRIES, default 8)
};

Arena& arena() {
    static Arena* a = new Arena;  // (leaked on purpose: tensors may be released after static destructors have run)
    return *a;
}

size_t round_up(size_t v, size_t to) { return (v + to - 1) / to * to; }

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

// physical memory back to the driver; the address range stays reserved (see the header: never map a range twice)
void release_block(Block* b) {
    if (b->va && b->bytes)
        for (size_t off = 0; off < b->bytes; off += b->chunk) (void)hipMemUnmap((char*)b->va + off, b->chunk);  // mapping by mapping
    for (auto h : b->chunks) (void)hipMemRelease(h);
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

// a new block of `bytes` (a multiple of the chunk size) on `device`, or nullptr (no memory / no virtual-memory support)
Block* create_block(Arena& a, int device, size_t bytes) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    Block* b = new Block;
    b->device = device;
    b->chunk = a.chunk;
    const size_t n = bytes / a.chunk;
    b->chunks.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, a.chunk, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            release_block(b);
            return nullptr;
        }
        b->chunks.push_back(h);
    }
    if (hipMemAddressReserve(&b->va, bytes, 0, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        b->va = nullptr;
        release_block(b);
        return nullptr;
    }
    for (size_t i = 0; i < n; ++i)
        if (hipMemMap((char*)b->va + i * a.chunk, a.chunk, 0, b->chunks[i], 0) != hipSuccess) {
            (void)hipGetLastError();
            for (size_t q = 0; q < i; ++q) (void)hipMemUnmap((char*)b->va + q * a.chunk, a.chunk);
            release_block(b);
            return nullptr;
        }
    b->bytes = bytes;  // (from here on release_block unmaps the whole range)
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(b->va, bytes, &acc, 1) != hipSuccess) {
        (void)hipGetLastError();
        release_block(b);
        return nullptr;
    }
    return b;
}

}  // namespace
}  // namespace cnsn

using namespace cnsn;

extern "C" {

void* cnsn_arena_alloc(int device, size_t bytes, void* stream) {
    if (bytes == 0 || device < 0) return nullptr;
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    if (a.broken) return nullptr;
    const size_t chunk = resolve_chunk(a, device);
    const size_t need = round_up(bytes, chunk);
    DeviceArena& d = a.dev[device];
    Block* b = nullptr;
    // a free block of exactly this many chunks: the fastest measured one, among equals one last used on the requesting stream
    auto range = d.free_by_size.equal_range(need);
    auto pick = range.second;
    for (auto it = range.first; it != range.second; ++it) {
        if (it->second->chunk != chunk) continue;  // (a block from before cnsn_arena_set_chunk_bytes)
        if (pick == range.second) {
            pick = it;
            continue;
        }
        const Block *have = pick->second, *cand = it->second;
        if (cand->gbps > have->gbps ||
            (cand->gbps == have->gbps && cand->stream == (hipStream_t)stream && have->stream != (hipStream_t)stream))
            pick = it;
    }
    if (pick != range.second) {
        b = pick->second;
        d.free_by_size.erase(pick);
        ++d.hits;
        if (b->stream != (hipStream_t)stream) {  // everything queued on the previous owner's stream so far comes first
            hipEvent_t ev;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
                (void)hipEventRecord(ev, b->stream);
                (void)hipStreamWaitEvent((hipStream_t)stream, ev, 0);
                (void)hipEventDestroy(ev);  // (released once the wait has been satisfied)
            } else {
                (void)hipGetLastError();
                (void)hipStreamSynchronize(b->stream);
            }
        }
    } else {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        // Best of `tries`: WHERE a block lies physically decides how fast it is written (about one candidate in five is of the
        // fast kind), so a new block of kTimedFrom bytes or more is chosen among a few candidates created together — distinct
        // physical memory —, each timed with the plane-strided fill (~1 ms), the losers' memory handed back at once.  Happens
        // when a block is CREATED (the first steps of a job), costs `tries` x the block's size transiently, never more than a
        // quarter of what is free.
        int tries = need >= kTimedFrom ? resolve_tries() : 1;
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                if (need > free_b) tries = 0;  // (no point in creating thousands of chunks to find that out)
                else if (tries > 1) tries = (int)std::max<size_t>(1, std::min<size_t>((size_t)tries, free_b / 4 / need));
            } else {
                (void)hipGetLastError();
            }
        }
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;  // (never time anything on a capturing stream)
        if (tries > 1 && (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)) {
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
            Block* c = create_block(a, device, need);
            if (!c) break;
            cand.push_back(c);
        }
        if (cand.size() > 1) {
            for (Block* c : cand) c->gbps = measure_block(c, (hipStream_t)stream);
            std::sort(cand.begin(), cand.end(), [](const Block* x, const Block* y) { return x->gbps > y->gbps; });
            // (keeping a fast runner-up for the next request of the size was tried: the second-best of eight is slower than the best
            // of the next eight — 0.766-0.772 against 0.759 ms per headline step on one box; not kept)
            for (size_t i = 1; i < cand.size(); ++i) release_block(cand[i]);  // (measure_block left the stream idle)
            d.probed += cand.size();
        }
        for (auto h : spacers) (void)hipMemRelease(h);
        b = cand.empty() ? nullptr : cand[0];
        if (cur != device && cur >= 0) (void)hipSetDevice(cur);
        if (!b) {
            ++d.failed;
            if (d.blocks == 0 && d.failed >= 2) a.broken = true;  // never worked on this system: stop asking the driver
            return nullptr;
        }
        ++d.misses;
        ++d.blocks;
        d.mapped += b->bytes;
        a.by_ptr[b->va] = b;
    }
    b->stream = (hipStream_t)stream;
    b->in_use = true;
    d.in_use += b->bytes;
    ++d.blocks_in_use;
    return b->va;
}

int cnsn_arena_free(void* ptr) {
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    auto it = a.by_ptr.find(ptr);
    if (it == a.by_ptr.end() || !it->second->in_use) return CNSN_E_NULL;
    Block* b = it->second;
    DeviceArena& d = a.dev[b->device];
    b->in_use = false;
    d.in_use -= b->bytes;
    --d.blocks_in_use;
    d.free_by_size.emplace(b->bytes, b);
    return CNSN_OK;
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
    for (Block* b : drop) {  // work queued on the block's stream may still touch it: settle that stream first
        (void)hipSetDevice(b->device);
        (void)hipStreamSynchronize(b->stream);
        freed += b->bytes;
        release_block(b);
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    return freed;
}

int cnsn_arena_prospect(int device, size_t bytes, int keep, int candidates, void* stream, float* gbps_out) {
    if (bytes == 0 || device < 0 || keep < 0 || candidates <= 0) return CNSN_E_NULL;
    Arena& a = arena();
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != device) (void)hipSetDevice(device);
    std::vector<Block*> made;
    size_t need = 0;
    {
        std::lock_guard<std::mutex> lock(a.mu);
        if (a.broken) {
            if (cur != device && cur >= 0) (void)hipSetDevice(cur);
            return 0;
        }
        const size_t chunk = resolve_chunk(a, device);
        need = round_up(bytes, chunk);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)  // never more than half of what is free
            candidates = (int)std::min<size_t>((size_t)candidates, free_b / 2 / need);
        else
            (void)hipGetLastError();
        for (int i = 0; i < candidates; ++i) {  // all alive at once: distinct physical memory
            Block* b = create_block(a, device, need);
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
            ++d.blocks;
            d.mapped += b->bytes;
            a.by_ptr[b->va] = b;
            d.free_by_size.emplace(b->bytes, b);
            ++kept;
        }
    }
    (void)hipStreamSynchronize((hipStream_t)stream);
    for (size_t i = (size_t)kept; i < order.size(); ++i) release_block(order[i]);
    if (cur != device && cur >= 0) (void)hipSetDevice(cur);
    return kept;
}

int cnsn_arena_block_gbps(const void* ptr, float* gbps) {
    if (!gbps) return CNSN_E_NULL;
    Arena& a = arena();
    std::lock_guard<std::mutex> lock(a.mu);
    auto it = a.by_ptr.find((void*)ptr);
    if (it == a.by_ptr.end()) return CNSN_E_NULL;
    *gbps = it->second->gbps;
    return CNSN_OK;
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
    a.broken = false;
    return CNSN_OK;
}

}  // extern "C"
