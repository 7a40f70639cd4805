// Channel-resident strategy: ONE launch per direction; every plane is read from HBM exactly once.
//
// Why it is possible: all coupling of the fused op — CrossNorm's batch permutation (cnsn.py:62,
// same channel) and SelfNorm's BatchNorm1d over N (cnsn.py:121,138) — stays inside one channel.
// A channel (N planes) is given to a CLUSTER of K co-resident workgroups; each wave keeps PPW
// whole planes in its VGPRs (NV 16-byte vectors per lane and plane), so
//     forward : read x once -> exact two-pass plane statistics from registers -> exchange N*NG
//               scalars inside the cluster -> BatchNorm/gate algebra -> apply from registers -> write y
//     backward: read G and x once -> per-plane sums -> exchange -> mid algebra -> apply -> write dx
// i.e. 2*E*b and 3*E*b bytes of HBM traffic instead of the two-pass strategy's 3*E*b and 5*E*b.
//
// Cluster exchange (MI355X_MICROARCH.md "handoff"/"allgather", cdna_hip_programming.md G16 R2):
// each plane's scalars are published two floats at a time as 8-byte granules, ONE relaxed agent-scope
// (sc1, write-through) atomic store each; one wave per workgroup re-reads the channel's granules with
// relaxed agent-scope loads until none holds the 'empty' pattern.  No fences, no flags, no counters;
// the granule area is memset to the empty pattern on the stream before every launch.
//
// Deadlock freedom: the grid is persistent, G = (resident workgroups) rounded down to a multiple
// of K, and workgroup b handles items b, b+G, ...; the K members of a cluster are therefore
// always the same K co-resident workgroups working on the same iteration.  Kernels of other streams
// (RCCL all-reduce, MIOpen) may delay residency but always finish, so waiting is safe.  Partial residency is
// not a deadlock either: workgroups are dispatched in block order (per XCD), a cluster is K consecutive blocks
// and waits only for its own members, so the lowest incomplete cluster always receives the next free slots
// while complete clusters run to their end.  Every spin is nevertheless bounded (seconds): on time-out the
// kernel raises a word in the control block (every other wait drains at once), counts the event in a
// host-visible word (cnsn_resident_timeouts(): the host learns without synchronising, stops using this
// strategy and reports the step as invalid) and RETURNS — no trap, the HIP context stays usable.
#pragma once
#include <type_traits>

#include "../../include/cnsn_hip.h"
#include "cnsn_algebra.h"
#include "cnsn_device.h"
#include "cnsn_layout.h"

#ifndef CNSN_POLL_SLEEP
#define CNSN_POLL_SLEEP 4  // s_sleep units (64 clocks) between two polls of the granules
#endif

namespace cnsn {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr long long kWaitLimitTicks = 5ll * 100000000ll;  // default: 5 s of the 100 MHz wall clock (s_memrealtime)
constexpr int kCtlBytes = 256;  // control block in front of the granule area (word 0: time-out flag)

struct ResArgs {
    MidArgs mid;
    int M, Wd, nvec;
    Box cb, sb;
    int K;      // workgroups per channel
    int items;  // C * K
    int stagger;  // start-up skew between clusters, in units of s_sleep(127) (~3.4 us)
    unsigned epoch;       // 0: granules are float PAIRS in a workspace memset to all-ones before the launch;
                          // > 0: ONE float + this tag per granule, in a persistent context that is never cleared
                          // (cnsn_context_init): a granule is present when its tag equals the launch's epoch
    unsigned ctl_idle;    // value of the control word while nobody has given up (all-ones after the memset, 0 in a context)
    unsigned* host_flag;  // pinned host word counting timed-out launches (NULL: not available)
    long long wait_ticks;  // bound of a cluster wait in 100 MHz ticks (default 5 s; CNSN_WAIT_MS)
    int fault;             // tests only (CNSN_FAULT_INJECT=1): the last member of channel 0's cluster never publishes
    unsigned long long* prof;  // tuning builds (-DCNSN_PROF): [workgroup < 64][iteration < 16][8] time stamps
    int xcd;               // 1: clusters are made of workgroups of ONE XCD (cluster_block; CNSN_XCD=1, off by default)
};

// XCD-aware cluster membership (round 5).  The dispatcher hands consecutive workgroups of a grid to the 8 XCDs round-robin
// (workgroup b runs on XCD b % 8, each XCD working through ITS workgroups in increasing order), and every XCD has its own L2.
// A cluster of K consecutive workgroups therefore spans all 8 XCDs: every granule a member publishes is read through seven
// other L2s, which the fabric keeps coherent at the price of a round trip past the Infinity Cache (~2 us under load) for each
// poll.  With this numbering — workgroup b is "virtual block" (b % 8) * G/8 + b / 8 — K consecutive virtual blocks are K
// consecutive workgroups of ONE XCD's queue: the cluster's exchange stays in one L2.  The in-order argument of DESIGN §5
// (a cluster waits only for members that were dispatched before or with it) holds per XCD queue.  Needs a grid that is a
// multiple of 8; clusters that straddle two XCDs (G/8 not a multiple of K) just lose the locality.
// MEASURED SLOWER (profiles/r05_xcd_clusters.md: fp32 forward 0.300 -> 0.375 ms, everything else 1-4 % slower): with a
// channel's 256 planes behind one L2 the chip's memory traffic of the moment goes through one XCD's fabric port per
// channel instead of all eight.  Kept as an A/B knob, off by default.
__device__ __forceinline__ int cluster_block(int xcd_on) {
    const unsigned b = blockIdx.x, G = gridDim.x;
    return (xcd_on && (G & 7u) == 0u) ? (int)((b & 7u) * (G >> 3) + (b >> 3)) : (int)b;
}

// The batch permutation as a LAUNCH ARGUMENT (cnsn_problem_t.perm_host, ABI 5): 16-bit indices in the kernarg segment, so
// that no host-to-device copy sits in front of the launch (models/cnsn.py:62 draws it on the host: torch.randperm(N) from
// the CPU generator).  on = 0: the kernel reads the device array `perm` as before.
constexpr int kPermInlineMax = 1024;  // = CNSN_PERM_INLINE_MAX
struct PermInline {
    int on;
    unsigned short v[kPermInlineMax];
};
__device__ __forceinline__ int perm_at(const int64_t* __restrict__ perm, const PermInline& pin, int n) {
    return pin.on ? (int)pin.v[n] : (int)perm[n];
}

#ifdef CNSN_PROF
#ifndef CNSN_PROF_SKIP
#define CNSN_PROF_SKIP 0  // first iteration recorded
#endif
#ifndef CNSN_PROF_WG0
#define CNSN_PROF_WG0 0   // first workgroup recorded
#endif
#define CNSN_STAMP(slot)                                                                        \
    do {                                                                                        \
        if (ra.prof && threadIdx.x == 0 && blockIdx.x >= CNSN_PROF_WG0 && blockIdx.x < CNSN_PROF_WG0 + 64 && iter_ >= CNSN_PROF_SKIP && iter_ < CNSN_PROF_SKIP + 16) \
            ra.prof[((size_t)(blockIdx.x - CNSN_PROF_WG0) * 16 + iter_ - CNSN_PROF_SKIP) * 8 + (slot)] = wall_clock64(); \
    } while (0)
#define CNSN_NOTE(slot, v)                                                          \
    do {                                                                           \
        if (ra.prof && threadIdx.x == 0 && blockIdx.x >= CNSN_PROF_WG0 && blockIdx.x < CNSN_PROF_WG0 + 64 && iter_ >= CNSN_PROF_SKIP && iter_ < CNSN_PROF_SKIP + 16) \
            ra.prof[((size_t)(blockIdx.x - CNSN_PROF_WG0) * 16 + iter_ - CNSN_PROF_SKIP) * 8 + (slot)] = (v);        \
    } while (0)
#else
#define CNSN_STAMP(slot) \
    do {                 \
    } while (0)
#define CNSN_NOTE(slot, v) \
    do {                   \
        (void)(v);         \
    } while (0)
#endif

// De-synchronise the clusters at start-up: every cluster runs the same load -> exchange -> store cycle
// with the same period, so clusters that start together stay in lock step and the whole chip idles
// its memory system during every exchange.  Skewing the start by a fraction of the period lets one
// cluster's exchange overlap another's loads and stores.
__device__ __forceinline__ void startup_skew(const ResArgs& ra) {
    const int steps = ((cluster_block(ra.xcd) / ra.K) & 3) * ra.stagger;
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(127);
}

// A granule is ONE naturally aligned 8-byte word holding TWO floats, written by one relaxed agent-scope
// (sc1, write-through) atomic store.  "Not yet written" is the all-ones pattern the granule area is
// memset to before every launch; NaNs are canonicalised on publish so a payload can never equal it.
constexpr unsigned long long kGranuleEmpty = ~0ull;
constexpr unsigned kCtlIdle = 0xffffffffu;  // control word 0 after the memset; anything else = give up

__device__ __forceinline__ unsigned canon_bits(float v) { return v != v ? 0x7fc00000u : __float_as_uint(v); }

__device__ __forceinline__ void put_granule(unsigned long long* p, float lo, float hi) {
    __hip_atomic_store((gu64*)p, ((unsigned long long)canon_bits(hi) << 32) | (unsigned long long)canon_bits(lo),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// tagged form (persistent context): one float + the launch's epoch per granule
__device__ __forceinline__ void put_tagged(unsigned long long* p, float v, unsigned epoch) {
    __hip_atomic_store((gu64*)p, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// ONE wave gathers `total` granules (2*total floats) into LDS; re-reads all of them until none is empty.
// Returns false when it gave up (time-out, or another workgroup already did): the caller's workgroup must leave.
// (epoch > 0: `total` granules of ONE float each, present when the tag matches — see ResArgs::epoch)
__device__ __forceinline__ bool sweep_granules(const unsigned long long* g, int total, float* vals, unsigned* ctl,
                                               unsigned* host_flag, long long wait_ticks, unsigned& passes, unsigned epoch,
                                               unsigned ctl_idle) {
    const int lane = threadIdx.x & 63;
    long long t_start = 0;
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
        if (epoch) {
            for (int i = lane; i < total; i += 64) {
                const unsigned long long v = __hip_atomic_load((gu64*)(g + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok &= (unsigned)(v >> 32) == epoch;
                vals[i] = __uint_as_float((unsigned)v);
            }
        } else {
            for (int i = lane; i < total; i += 64) {
                const unsigned long long v = __hip_atomic_load((gu64*)(g + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok &= v != kGranuleEmpty;
                vals[2 * i] = __uint_as_float((unsigned)v);
                vals[2 * i + 1] = __uint_as_float((unsigned)(v >> 32));
            }
        }
        passes = spins + 1;
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(CNSN_POLL_SLEEP);
        if ((spins & 15u) == 15u) {
            const long long now = (long long)wall_clock64();
            if (t_start == 0) t_start = now;
            const unsigned seen = __hip_atomic_load((gu32*)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen != ctl_idle) return false;  // somebody gave up already: drain
            if (now - t_start > wait_ticks) {
                // A cluster member never published: part of the grid was kept off the device for seconds (the GPU
                // is shared with something that never yields).  The FIRST workgroup to notice flips the control
                // word (every other wait drains), bumps the host-visible counter, and everybody returns: the
                // outputs of this launch are incomplete, which the host learns from cnsn_resident_timeouts().
                if (lane == 0) {
                    const unsigned prev = __hip_atomic_exchange((gu32*)ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (prev == ctl_idle && host_flag)
                        __hip_atomic_fetch_add(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                return false;
            }
        }
    }
}

// ---- the same gather through the SCALAR memory path --------------------------------------------------------
// Why: a vector load issued while the CU's other waves have ~200 KB of plane loads in flight returns only after
// them (the vector L1 returns in order): ~9 us per poll at the north-star shape, and a cluster wait needs 2-3
// polls (profiles/r01_resident_tuning.md).  Scalar loads take the other road — scalar data cache -> L2, out of
// order, `glc` = never served from the scalar cache — so a poll costs an L2 round trip.  ALL FOUR waves take
// part: wave w owns the 256-byte groups w, w+4, ... of the channel's granules, loads a group with four
// s_load_dwordx16 into 64 SGPRs, moves them to one VGPR (v_writelane), stores that to LDS and tests it for the
// 'empty' pattern; only groups that still had empties are read again.  A granule is an aligned 8-byte word written
// by one store, so both of its dwords are empty or neither is: the test is per dword.
#ifndef CNSN_SWEEP_SCALAR
#define CNSN_SWEEP_SCALAR 0  // measured (profiles/r02_resident_phases.md): last publish -> seen 3.2 us instead of 5.5 at
                             // the north-star shape, but no faster end to end and slower for one-item grids
#endif
// critical-path experiments on the exchange -> algebra -> apply section (round 2, profiles/r02_resident_phases.md).  Each
// was measured on its own against the plain build; none pays, so all are OFF — kept because the measurements are
// the documentation of where the time does NOT go:
#ifndef CNSN_PRIO
#define CNSN_PRIO 0        // wave priorities by phase (2 load/sum/publish, 0 while polling, 3 algebra/apply): +-1 % at the
#endif                     // north-star shape, and 2x SLOWER for one-item grids ((128,32,32,32) backward 0.043 -> 0.109 ms)
#ifndef CNSN_CHEAP_SHIFT
#define CNSN_CHEAP_SHIFT 0 // batch sums about a shift taken from plane 0's raw moments instead of its full algebra: no change
#endif
#ifndef CNSN_WAVE_COEF
#define CNSN_WAVE_COEF 0   // coefficients of a wave's planes computed by its own lanes and handed over with v_readlane
#endif                     // (no LDS round trip, one barrier less): algebra phase 4.4 -> 2.9 us, kernel time unchanged
                           // (fp32) to 7 % slower (bf16, two planes per wave)
typedef unsigned su16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned sload_glc_u32(const void* p) {
    unsigned v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// lanes BASE .. BASE+15 of v <- the 16 dwords of r (v_writelane_b32; this clang has no builtin for it)
template <int BASE>
__device__ __forceinline__ void put16(unsigned& v, const su16_t& r) {
#define CNSN_WL(i) asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(r[i]), "n"(BASE + (i)))
    CNSN_WL(0); CNSN_WL(1); CNSN_WL(2); CNSN_WL(3); CNSN_WL(4); CNSN_WL(5); CNSN_WL(6); CNSN_WL(7);
    CNSN_WL(8); CNSN_WL(9); CNSN_WL(10); CNSN_WL(11); CNSN_WL(12); CNSN_WL(13); CNSN_WL(14); CNSN_WL(15);
#undef CNSN_WL
}

// 64 dwords at base + off (bytes) -> lane l holds dword l
__device__ __forceinline__ unsigned sload_group(const void* base, unsigned off) {
    su16_t a, b, c, d;
    asm volatile(
        "s_load_dwordx16 %0, %4, %5 glc\n\t"
        "s_load_dwordx16 %1, %4, %6 glc\n\t"
        "s_load_dwordx16 %2, %4, %7 glc\n\t"
        "s_load_dwordx16 %3, %4, %8 glc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d)
        : "s"(base), "s"(off), "s"(off + 64u), "s"(off + 128u), "s"(off + 192u)
        : "memory");
    unsigned v = 0;
    put16<0>(v, a);
    put16<16>(v, b);
    put16<32>(v, c);
    put16<48>(v, d);
    return v;
}

// Every wave of the workgroup calls this (wave-uniform arguments); returns false when this wave gave up.
// nfloats = 2 * granules of the channel; vals must hold nfloats floats.
__device__ __forceinline__ bool sweep_granules_scalar(const unsigned long long* g, int nfloats, float* vals, unsigned* ctl,
                                                      unsigned* host_flag, long long wait_ticks, int wave,
                                                      unsigned& passes) {
    const int lane = threadIdx.x & 63;
    const int ngroups = (nfloats + 63) >> 6;
    const int mine = ngroups > wave ? (ngroups - wave + 3) >> 2 : 0;  // groups wave, wave+4, ...
    unsigned long long todo = mine >= 64 ? ~0ull : ((1ull << mine) - 1ull);  // (more than 64: all re-read each pass)
    long long t_start = 0;
    for (unsigned spins = 0;; ++spins) {
        bool all = true;
        for (int k = 0; k < mine; ++k) {
            if (k < 64 && !((todo >> k) & 1ull)) continue;
            const int grp = wave + 4 * k;
            const unsigned v = sload_group((const void*)g, (unsigned)grp * 256u);
            const int idx = grp * 64 + lane;
            const bool valid = idx < nfloats;
            if (valid) vals[idx] = __uint_as_float(v);
            const bool ok = __ballot(valid && v == 0xffffffffu) == 0ull;
            if (ok && k < 64) todo &= ~(1ull << k);
            all &= ok;
        }
        passes = spins + 1;
        if (all) return true;
        __builtin_amdgcn_s_sleep(2);
        if ((spins & 15u) == 15u) {
            const long long now = (long long)wall_clock64();
            if (t_start == 0) t_start = now;
            if (sload_glc_u32(ctl) != kCtlIdle) return false;  // somebody gave up already: drain (untagged workspace only)
            if (now - t_start > wait_ticks) {
                if (lane == 0) {
                    const unsigned prev = __hip_atomic_exchange((gu32*)ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (prev == kCtlIdle && host_flag)
                        __hip_atomic_fetch_add(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                return false;
            }
        }
    }
}

__device__ __forceinline__ float keep_if(float v, int mask) { return __int_as_float(__float_as_int(v) & mask); }
// a or b by an all-ones / all-zeros word, branch-free: (a & m) | (b & ~m) is one v_bfi_b32.  (`in_box ? f(x) : g(x)` with a
// per-element predicate was compiled into an exec-masked if / else per ELEMENT — s_and_b64 / s_xor_b64 / s_mov_b64 exec /
// s_andn2_saveexec / s_or_b64 around three instructions a branch, the 112 loop-invariant lane masks of a plane's elements kept in
// scalar register pairs and spilled to vector lanes: two v_readlane per element — round 4, ISA of the boxed kernels.)
__device__ __forceinline__ float pick_if(int mask, float a, float b) {
    float r;  // spelled out: the compiler turns the C expression back into a compare + v_cndmask and hoists the compares
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}
// bit `pos` of `word` as an all-ones / all-zeros word (one v_bfe_i32), opaque to the optimiser for the same reason
__device__ __forceinline__ int bit_mask(unsigned word, int pos) {
#if defined(CNSN_FAKE_MASKS)  // timing experiment only (WRONG results): what the kernels would gain if an element's mask cost nothing
    (void)pos;
    return (int)word;
#endif
    int r = (int)(word << (31 - pos)) >> 31;
    asm("" : "+v"(r));
    return r;
}

// static per-lane geometry of the register slots: slot j of this lane holds vector j*64+lane.
// Kept small on purpose (the planes themselves want the registers): validity is one compare against
// nvec, box membership is one bit per element packed into (NV*VEC+31)/32 words.
template <int VEC, int NV, bool BOXED>
struct SlotGeom {
    static constexpr int WORDS = BOXED ? (NV * VEC + 31) / 32 : 1;
    int lane, nvec;  // (SPLIT kernels: lane carries the wave's slot offset, lane + 64 * first slot of the wave)
    int shift;       // slots are RIGHT-aligned by this many (PlaneIo's layout: slot j holds vector (j - shift) * 64 + lane); 0 = left-aligned
    mutable unsigned cbits[WORDS];  // bit j*VEC+q: element q of slot j is inside the content box
    mutable unsigned sbits[WORDS];  // ... inside the style box   (mutable: forget() below)
    __device__ __forceinline__ SlotGeom(const ResArgs& ra, int lane_, int shift_ = 0) : lane(lane_), nvec(ra.nvec), shift(shift_) {
#pragma unroll
        for (int w = 0; w < WORDS; ++w) cbits[w] = sbits[w] = 0;
        if constexpr (BOXED) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = (j - shift) * 64 + lane;
                if (j >= shift && i < ra.nvec) {
                    const int e = i * VEC;  // VEC divides the width: one row per vector
                    const int r = e / ra.Wd, c = e - r * ra.Wd;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const int p = j * VEC + q;
                        cbits[p >> 5] |= (ra.cb.has(r, c + q) ? 1u : 0u) << (p & 31);
                        sbits[p >> 5] |= (ra.sb.has(r, c + q) ? 1u : 0u) << (p & 31);
                    }
                }
            }
        }
    }
    __device__ __forceinline__ bool valid(int j) const { return j >= shift && (j - shift) * 64 + lane < nvec; }
    __device__ __forceinline__ bool in_c(int j, int q) const {
        const int p = j * VEC + q;
        return (cbits[p >> 5] & (1u << (p & 31))) != 0u;
    }
    __device__ __forceinline__ bool in_s(int j, int q) const {
        const int p = j * VEC + q;
        return (sbits[p >> 5] & (1u << (p & 31))) != 0u;
    }
    // the same memberships as all-ones / all-zeros words (one v_bfe_i32): keep_if() keeps or zeroes a float by one, pick_if()
    // chooses between two
    __device__ __forceinline__ int mask_c(int j, int q) const {
        const int p = j * VEC + q;
        return bit_mask(cbits[p >> 5], p & 31);
    }
    __device__ __forceinline__ int mask_s(int j, int q) const {
        const int p = j * VEC + q;
        return bit_mask(sbits[p >> 5], p & 31);
    }
    // The bit words never change, so everything derived from them is loop-invariant and would be hoisted out of the item
    // loop: 2 x NV x VEC mask words (or lane-mask register pairs) per lane, spilled.  Called at the start of every per-plane
    // loop that uses the masks, this makes the optimiser forget what the words hold, and an element's mask is extracted where it
    // is used.
    __device__ __forceinline__ void forget() const {
        if constexpr (BOXED) {
#pragma unroll
            for (int w = 0; w < WORDS; ++w) asm volatile("" : "+v"(cbits[w]), "+v"(sbits[w]));
        }
    }
};

// Plane access through buffer instructions.  Every register slot j of a plane gets its own resource
// descriptor (4 SGPRs, scalar ALU only): base = plane + j*64 vectors, num_records = bytes of the plane
// left from there (clamped at 0).  All slots then share ONE per-lane VGPR offset (lane * vector bytes),
// and the hardware range check (VGPR offset >= num_records) makes lanes past the end of the plane —
// and whole planes past the end of the batch — read zeros and drop their stores.  No address VGPRs.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));

template <int BYTES>
struct RawOf;
template <>
struct RawOf<16> {
    using type = v4i_t;
};
template <>
struct RawOf<8> {
    using type = v2i_t;
};
template <typename T, int VEC>
using Raw = typename RawOf<(int)sizeof(T) * VEC>::type;  // VEC elements of T as 32-bit words

// element q of a raw vector as float / raw vector from VEC floats (no aggregate bit-casts: handing a
// bit-cast struct to the buffer builtins was miscompiled by ROCm 7.2 hipcc in these kernels)
template <typename T, int VEC>
__device__ __forceinline__ float elem(const Raw<T, VEC>& r, int q) {
    if constexpr (sizeof(T) == 4) {
        return __int_as_float(r[q]);
    } else {
        const unsigned w = (unsigned)r[q >> 1];
        const unsigned short h = (q & 1) ? (unsigned short)(w >> 16) : (unsigned short)(w & 0xffffu);
        if constexpr (__is_same(T, bf16_t))
            return __uint_as_float((unsigned)h << 16);
        else
            return (float)__builtin_bit_cast(_Float16, h);
    }
}
// 16-bit element types: acc + a.lo*b.lo + a.hi*b.hi straight from two packed 32-bit words (v_dot2c_f32_bf16 / v_dot2c_f32_f16:
// fp32 accumulation of exact products, NO unpacking) — a bandwidth-bound kernel on MI355X has ~20-30 vector instructions per
// 16-bit element to spend, and turning every element into a float first (shift / mask, then add, subtract, multiply-add) took
// 5 of them per element and tensor in the statistics loops (ISA count, round 4: profiles/r04_dot2_sums.md)
#ifndef CNSN_DOT2
#define CNSN_DOT2 1
#endif
typedef __attribute__((ext_vector_type(2))) __bf16 cnsn_bf2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 cnsn_h2_t;
typedef __attribute__((ext_vector_type(2))) float cnsn_f2_t;
template <typename T>
__device__ __forceinline__ float dot2_acc(unsigned a, unsigned b, float acc) {
    static_assert(sizeof(T) == 2, "packed 16-bit words only");
    if constexpr (__is_same(T, bf16_t))
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(cnsn_bf2_t, a), __builtin_bit_cast(cnsn_bf2_t, b), acc, false);
    else
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(cnsn_h2_t, a), __builtin_bit_cast(cnsn_h2_t, b), acc, false);
}
template <typename T>
__device__ __forceinline__ constexpr unsigned ones2() {  // (1, 1) as a packed word
    return __is_same(T, bf16_t) ? 0x3f803f80u : 0x3c003c00u;
}
// the two elements of a packed word as a float pair (what v_pk_add_f32 / v_pk_fma_f32 work on)
template <typename T>
__device__ __forceinline__ cnsn_f2_t unpack2(unsigned w) {
    cnsn_f2_t r;
    if constexpr (__is_same(T, bf16_t)) {
        r.x = __uint_as_float(w << 16);
        r.y = __uint_as_float(w & 0xffff0000u);
    } else {
        const cnsn_h2_t h = __builtin_bit_cast(cnsn_h2_t, w);
        r.x = (float)h.x;
        r.y = (float)h.y;
    }
    return r;
}

// two floats -> one 32-bit word of two 16-bit elements (lo in bits 0..15), round to nearest even.  Converting the PAIR as
// a vector lets the compiler pair the arithmetic that feeds it the same way (v_pk_fma_f32 on {lo, hi}) and finish with one
// v_cvt_pk_bf16_f32 — converting element by element made it pair the even elements of two words and re-interleave the
// halves afterwards (v_and / v_lshl / 2 v_or_sdwa per word: a third of the apply loops' vector instructions).
typedef float v2f_t __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ int pack2(float lo, float hi) {
    static_assert(sizeof(T) == 2, "16-bit element types");
    const v2f_t p = {lo, hi};
    if constexpr (__is_same(T, bf16_t)) {
        typedef __bf16 v2bf_t __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(int, __builtin_convertvector(p, v2bf_t));
    } else {
        typedef _Float16 v2h_t __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(int, __builtin_convertvector(p, v2h_t));
    }
}
template <typename T, int VEC>
__device__ __forceinline__ Raw<T, VEC> pack(const float (&f)[VEC]) {
    Raw<T, VEC> r;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) r[q] = __float_as_int(f[q]);
    } else {
#pragma unroll
        for (int q = 0; q < VEC; q += 2) r[q >> 1] = pack2<T>(f[q], f[q + 1]);
    }
    return r;
}

// ---- element PAIRS (v_pk_add_f32 / v_pk_fma_f32 do two floats an instruction at the rate of one: with crop boxes the cluster
// kernels are bound by the vector ALU, not by HBM — 13-20 instructions per element — so the boxed statistics and apply loops
// work on pairs; the masks stay per element (one v_bfe_i32 each))
template <typename T, int VEC>
__device__ __forceinline__ cnsn_f2_t elem2(const Raw<T, VEC>& r, int p) {  // elements 2p, 2p+1
    if constexpr (sizeof(T) == 4) {
        const cnsn_f2_t v = {__int_as_float(r[2 * p]), __int_as_float(r[2 * p + 1])};
        return v;
    } else {
        return unpack2<T>((unsigned)r[p]);
    }
}
__device__ __forceinline__ cnsn_f2_t splat2(float v) {
    const cnsn_f2_t r = {v, v};
    return r;
}
__device__ __forceinline__ cnsn_f2_t keep_if2(cnsn_f2_t v, int m0, int m1) {
    const cnsn_f2_t r = {keep_if(v.x, m0), keep_if(v.y, m1)};
    return r;
}
__device__ __forceinline__ cnsn_f2_t pick_if2(int m0, int m1, cnsn_f2_t a, cnsn_f2_t b) {
    const cnsn_f2_t r = {pick_if(m0, a.x, b.x), pick_if(m1, a.y, b.y)};
    return r;
}
__device__ __forceinline__ cnsn_f2_t fma2(cnsn_f2_t a, cnsn_f2_t b, cnsn_f2_t c) { return __builtin_elementwise_fma(a, b, c); }
// pairs -> a raw vector (16-bit: one v_cvt_pk per pair)
template <typename T, int VEC>
__device__ __forceinline__ Raw<T, VEC> pack_pairs(const cnsn_f2_t (&f)[VEC / 2]) {
    Raw<T, VEC> r;
#pragma unroll
    for (int p = 0; p < VEC / 2; ++p) {
        if constexpr (sizeof(T) == 4) {
            r[2 * p] = __float_as_int(f[p].x);
            r[2 * p + 1] = __float_as_int(f[p].y);
        } else {
            r[p] = pack2<T>(f[p].x, f[p].y);
        }
    }
    return r;
}
// the plain sum of a plane's register slots (slots past the plane's end hold zeros); 16 bits: straight from the packed words
template <typename T, int VEC, int NV>
__device__ __forceinline__ float slots_sum(const Raw<T, VEC> (&d)[NV]) {
    float s0 = 0.f;
    if constexpr (CNSN_DOT2 && sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int w = 0; w < VEC / 2; ++w) s0 = dot2_acc<T>((unsigned)d[j][w], ones2<T>(), s0);
    } else {
        cnsn_f2_t s2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int q = 0; q < VEC / 2; ++q) s2 += elem2<T, VEC>(d[j], q);
        s0 = s2.x + s2.y;
    }
    return s0;
}
// the six region sums boxed_moments() takes (whole plane, content box, style box: sum d, sum d*d with d = x - k), this lane's part
template <typename T, int VEC, int NV, typename SG>
__device__ __forceinline__ void boxed_region_sums(const Raw<T, VEC> (&d)[NV], const SG& sg, float k, float (&t)[6]) {
    cnsn_f2_t a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0;
    const cnsn_f2_t k2 = splat2(k);
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (sg.valid(j)) {
#pragma unroll
            for (int q = 0; q < VEC / 2; ++q) {
                const cnsn_f2_t dd = elem2<T, VEC>(d[j], q) - k2;
                const cnsn_f2_t dc = keep_if2(dd, sg.mask_c(j, 2 * q), sg.mask_c(j, 2 * q + 1));
                const cnsn_f2_t ds = keep_if2(dd, sg.mask_s(j, 2 * q), sg.mask_s(j, 2 * q + 1));
                a0 += dd;
                a1 = fma2(dd, dd, a1);
                a2 += dc;
                a3 = fma2(dc, dc, a3);
                a4 += ds;
                a5 = fma2(ds, ds, a5);
            }
        }
    t[0] = a0.x + a0.y, t[1] = a1.x + a1.y, t[2] = a2.x + a2.y, t[3] = a3.x + a3.y, t[4] = a4.x + a4.y, t[5] = a5.x + a5.y;
}

// the four backward sums of a plane with crop boxes, this lane's part: sum G and sum G*(X - si) inside the content box (acc[0..1])
// and over the whole plane (acc[2..3]), both about the same shift si; valid(j) / mask_c(j, q) as SlotGeom's
template <typename T, int VEC, int NV, typename VALID, typename MASK>
__device__ __forceinline__ void boxed_bwd_sums(const Raw<T, VEC> (&g)[NV], const Raw<T, VEC> (&x)[NV], VALID valid, MASK mask_c, float si,
                                               float (&acc)[4]) {
    cnsn_f2_t a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    const cnsn_f2_t si2 = splat2(si);
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (valid(j)) {
#pragma unroll
            for (int q = 0; q < VEC / 2; ++q) {
                const cnsn_f2_t G = elem2<T, VEC>(g[j], q), Xc = elem2<T, VEC>(x[j], q) - si2;
                const cnsn_f2_t Gc = keep_if2(G, mask_c(j, 2 * q), mask_c(j, 2 * q + 1));
                a0 += Gc;
                a1 = fma2(Gc, Xc, a1);
                a2 += G;
                a3 = fma2(G, Xc, a3);
            }
        }
    acc[0] = a0.x + a0.y, acc[1] = a1.x + a1.y, acc[2] = a2.x + a2.y, acc[3] = a3.x + a3.y;
}

// Region moments of a plane with crop boxes from ONE masked pass about a common shift k (the plane mean, from a cheap
// unmasked pass before): sums of d = x - k and d*d over the whole plane, inside the content box and inside the style box.
// The outside-the-content-box moments follow by subtraction.  (The first version made two masked passes with three
// selects each — 26 vector instructions per element against 5 without boxes, which made the boxed kernels VALU-bound.)
//   in: St, Qt, Sc, Qc, Ss, Qs (wave-uniform sums), k, region sizes      out: pub[6] = mu_c, M2c, mu_o, M2o, mu_s, M2s
__device__ __forceinline__ void boxed_moments(float k, const float (&t)[6], int M, int Mc, int Ms, float (&pub)[6]) {
    const float St = t[0], Qt = t[1], Sc = t[2], Qc = t[3], Ss = t[4], Qs = t[5];
    const int Mo = M - Mc;
    const float rc = 1.f / (float)Mc, rs = 1.f / (float)Ms;
    pub[0] = k + Sc * rc;
    pub[1] = fmaxf(Qc - Sc * Sc * rc, 0.f);
    const float So = St - Sc, Qo = Qt - Qc;
    const float ro = Mo > 0 ? 1.f / (float)Mo : 0.f;
    pub[2] = Mo > 0 ? k + So * ro : 0.f;
    pub[3] = Mo > 0 ? fmaxf(Qo - So * So * ro, 0.f) : 0.f;
    pub[4] = k + Ss * rs;
    pub[5] = fmaxf(Qs - Ss * Ss * rs, 0.f);
}

// descriptor of slot j of the plane starting at `base` (plane_bytes = 0 for planes past the batch end)
template <typename T, int VEC>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slot_rsrc(const T* base, int plane_bytes, int j) {
    constexpr int SLOT = 64 * VEC * (int)sizeof(T);
    const int left = plane_bytes - j * SLOT;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)j * 64 * VEC), 0, left > 0 ? left : 0, 0x00020000);
}
// cache policy of the plane traffic: aux bit 1 = nt (non-temporal) — each line is used once.  Measured at
// (256,256,56,56) fp32: nt on loads and stores 0.921 ms/step vs 0.971 with the default policy.
#ifndef CNSN_RES_LOAD_AUX
#define CNSN_RES_LOAD_AUX 2
#endif
#ifndef CNSN_RES_STORE_AUX
#define CNSN_RES_STORE_AUX 2
#endif
template <typename T, int VEC>
__device__ __forceinline__ Raw<T, VEC> buf_load(__amdgpu_buffer_rsrc_t r, int voff) {
    if constexpr (sizeof(T) * VEC == 16)
        return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, CNSN_RES_LOAD_AUX);
    else
        return __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, CNSN_RES_LOAD_AUX);
}
template <typename T, int VEC>
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, int voff, const Raw<T, VEC>& v) {
    if constexpr (sizeof(T) * VEC == 16)
        __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, CNSN_RES_STORE_AUX);
    else
        __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, 0, CNSN_RES_STORE_AUX);
}

// the launch gave up (a bounded wait ran out): make the incomplete outputs loud — a NaN in the first vector of every
// plane this workgroup still owed (the host additionally sees cnsn_resident_timeouts() go up)
template <typename T, int VEC>
__device__ __forceinline__ void poison_plane(T* plane) {
    float f[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) f[q] = __builtin_nanf("");
    *reinterpret_cast<Raw<T, VEC>*>(plane) = pack<T, VEC>(f);
}

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// value of lane `src` (a compile-time constant after unrolling) in every lane
__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// Register budget: the planes a wave holds dominate its VGPR use, and the number of resident waves
// per SIMD (= workgroups per CU, a workgroup being one wave per SIMD) decides how much of the sync
// latency other workgroups can cover.  min-waves-per-SIMD handed to __launch_bounds__:
constexpr int data_regs(int elem_bytes, int vec, int nv, int ppw) { return ppw * nv * vec * elem_bytes / 4; }
#if defined(CNSN_WF) && defined(CNSN_WB)  // tuning builds: force the bounds
constexpr int fwd_waves(int, int = 4) { return CNSN_WF; }
constexpr int bwd_waves(int) { return CNSN_WB; }
#else
// measured on MI355X at (256,256,56,56) fp32/bf16 (profiles/r01_resident_tuning.md): forward 4, backward 2;
// tighter bounds make the compiler spill and every variant got slower.
// (16-bit planes held two per wave — 56+ data registers plus the unpacked floats of the vector being worked on —
//  spill at 128 VGPRs: 3 waves there; measured 0.204 vs 0.225 ms on the (256,256,56,56) bf16 forward)
constexpr int fwd_waves(int regs, int elem_bytes = 4) { return (elem_bytes == 2 && regs >= 56) ? 3 : 4; }
constexpr int bwd_waves(int) { return 3; }
#endif
// with the residual-block epilogue a second / third tensor is in flight next to the planes held.  Measured per
// instantiation class on the block shapes of ResNet-50 (profiles/r01_fused_block.md, "occupancy bounds"):
//   forward : 4 waves/SIMD, except 16-bit planes in 16-byte vectors (unpacking eight values per vector costs
//             registers): 3;   backward: 3, except the 16-bit 16-byte-vector planes of 7+ slots (56x56): 2.
#if defined(CNSN_WFE) && defined(CNSN_WBE)  // tuning builds
constexpr int fwd_waves_epi(int, int, int) { return CNSN_WFE; }
constexpr int bwd_waves_epi(int, int, int) { return CNSN_WBE; }
#else
constexpr int fwd_waves_epi(int elem_bytes, int vec, int) { return (elem_bytes == 2 && vec == 8) ? 3 : 4; }
constexpr int bwd_waves_epi(int elem_bytes, int vec, int nv) { return (elem_bytes == 2 && vec == 8 && nv >= 7) ? 2 : 3; }
#endif

// residual-block epilogue of the resident kernels (template flag EPI): the op's input is x + addend when
// `addend` is not NULL (rounded to T like the reference's `out += identity`), ReLU on the way out when `relu`
template <typename T>
__device__ __forceinline__ bool relu_open_r(float t) {
    if constexpr (sizeof(T) == 4)
        return t > 0.f;
    else
        return to_float(from_float<T>(t)) > 0.f;
}
template <typename T, int VEC>
__device__ __forceinline__ Raw<T, VEC> add_raw(const Raw<T, VEC>& a, const Raw<T, VEC>& b) {
    float f[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) f[q] = elem<T, VEC>(a, q) + elem<T, VEC>(b, q);
    return pack<T, VEC>(f);
}

// SPLIT kernels (a plane spread over the four waves of its workgroup): add wave-uniform partials across the waves.
// xch: [4][8] floats.  Every wave calls it; one workgroup barrier.
template <int NVAL>
__device__ __forceinline__ void split_merge(float (&v)[NVAL], float* xch, int wave, int lane) {
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NVAL; ++i) xch[wave * 8 + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NVAL; ++i) v[i] = (xch[i] + xch[8 + i]) + (xch[16 + i] + xch[24 + i]);
}

// dynamic LDS carve (bytes); NG = granules per plane, OWN = planes per workgroup
// rows of `saved` the backward algebra needs, staged in LDS per item (floats / doubles per instance)
enum StageF { F_A1 = 0, F_M_IN, F_MU_O, F_MU_P, F_G, F_F, F_A, F_SIG_P, F_SIG_C, F_M2C, F_SIG_S, F_N };
enum StageD { D_MU_C = 0, D_MU_S, D_ZH_G, D_ZH_F, D_N };

__host__ __device__ inline size_t res_lds_bytes(int N, int NG, int OWN, int coef_rows, bool backward) {
    return align16((size_t)N * NG * 4)        // vals[N][NG]
           + align16((size_t)2 * N * 8)       // dt [2][N]
           + align16((size_t)N * 4)           // perm / inverse perm [N]
           + align16((size_t)OWN * coef_rows * 4)  // coefficients of the owned planes
           + 4 * 4 * 8                        // block reduction scratch
           + 16                               // "this workgroup gave up" flag
           + 2 * 4 * 8 * 4                    // SPLIT kernels: per-wave partial statistics, two alternating copies
           + (backward ? align16((size_t)N * D_N * 8) + align16((size_t)N * F_N * 4) : 0);  // staged `saved`
}

// Granule regions without a fill launch (cnsn_resident.hip, resident_pong_*): the exchange of a launch runs through one of two
// untagged regions at the end of the persistent context; every workgroup first stores 'empty' (all ones) over its share of the
// OTHER region — control block and granules of the same extent — so that the next launch finds it clean.  The stores are
// ordered before that launch by the kernel boundary; nobody reads the other region during this launch.  clear_n = 0: nothing.
__device__ __forceinline__ void pipe_clear_other_region(unsigned long long* __restrict__ clear, unsigned clear_n) {
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < clear_n; i += gridDim.x * kBlock) clear[i] = ~0ull;
}

// ================================================================================================
// forward
// ================================================================================================
// SOLO (un-boxed only): nothing couples the planes of a channel — SelfNorm in eval mode (running statistics), no
// CrossNorm: the inference forward.  Every wave finishes its planes on its own: no publish, no cluster wait, no
// workgroup barrier; 2*E*b bytes at streaming speed.
// POST (with EPI, un-boxed): the addend joins AFTER the op — y = act(CNSN(x) + addend), the 'residual' / 'identity'
// positions of the callers (resnet_cnsn.py:112-116).  Its planes are fetched when the exchange is over (they are in
// flight during the algebra) and never held across the cluster wait; one workgroup per CU less than the other variants.
// SPLIT: ONE plane per workgroup, a quarter per wave (slots wave*NV .. wave*NV+NV-1): planes of 1025..4096 vectors
// (128x128 fp32).  The waves' partial sums meet in LDS (exact two-pass kept: the second pass runs about the merged
// means), wave 0 publishes; everything after the exchange is the same code with one plane per workgroup.
template <typename T, int VEC, int NV, int PPW, bool BOXED, bool EPI = false, bool SOLO = false, bool POST = false,
          bool SPLIT = false>
__global__ __launch_bounds__(kBlock, SPLIT ? (POST ? 2 : 3) : POST ? (data_regs(sizeof(T), VEC, NV, PPW) >= 64 ? 2 : 3) : EPI ? fwd_waves_epi((int)sizeof(T), VEC, NV) : fwd_waves(data_regs(sizeof(T), VEC, NV, PPW), (int)sizeof(T))) void resident_fwd_kernel(ResArgs ra, const T* __restrict__ x, T* __restrict__ y,
                                                              const int64_t* __restrict__ perm, GateDev gg, GateDev gf,
                                                              unsigned long long* __restrict__ gran,
                                                              double* __restrict__ saved, unsigned* __restrict__ ctl,
                                                              const T* __restrict__ addend, int relu,
                                                              unsigned long long* __restrict__ clear, unsigned clear_n,
                                                              PermInline pin) {
    pipe_clear_other_region(clear, clear_n);
    constexpr int NG = BOXED ? 6 : 2;
    constexpr int OWN = SPLIT ? 1 : 4 * PPW;
    static_assert(!SPLIT || (PPW == 1 && !CNSN_WAVE_COEF), "a split plane is the workgroup's only plane");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MidArgs a = ra.mid;
    const int N = a.N, C = a.C;
    float* vals = (float*)smem;
    double* zbuf = (double*)(smem + align16((size_t)N * NG * 4));
    int* sperm = (int*)((char*)zbuf + align16((size_t)2 * N * 8));
    float* ocoef = (float*)((char*)sperm + align16((size_t)N * 4));
    double* red = (double*)((char*)ocoef + align16((size_t)OWN * FC_ROWS * 4));
    int* gave_up = (int*)(red + 4 * 4);
    float* xch = (float*)(gave_up + 4);  // [2][4][8] (SPLIT)
    (void)zbuf;
    (void)xch;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: plane bases stay in SGPRs
    const size_t P = (size_t)N * C;
    const int j0 = SPLIT ? wave * NV : 0;  // first slot of this wave within its plane
    const SlotGeom<VEC, NV, BOXED> sg(ra, lane + 64 * j0);
    constexpr int VB = VEC * (int)sizeof(T);  // bytes per vector
    const int voff = lane * VB;

    if (a.cn_active)
        for (int n = threadIdx.x; n < N; n += kBlock) sperm[n] = perm_at(perm, pin, n);
    startup_skew(ra);
#if CNSN_PRIO
    __builtin_amdgcn_s_setprio(2);
#endif

    int iter_ = -1;
    for (int item = cluster_block(ra.xcd); item < ra.items; item += gridDim.x) {
        const int c = item / ra.K, k = item - c * ra.K;
        const int n0 = SPLIT ? k : (k * 4 + wave) * PPW;
        ++iter_;
        CNSN_STAMP(0);

        // ---- per-channel parameters: fetched first, ahead of the bulk loads in this CU's memory queue
        //      (a load issued in the algebra phase would wait behind every other workgroup's planes)
        float pw[4] = {0.f, 0.f, 0.f, 0.f}, pgam[2] = {0.f, 0.f}, pbet[2] = {0.f, 0.f}, prm[2] = {0.f, 0.f},
              prv[2] = {1.f, 1.f};
        if (a.sn_active) {
            pw[0] = gg.w[2 * c];
            pw[1] = gg.w[2 * c + 1];
            pgam[0] = gg.gamma[c];
            pbet[0] = gg.beta[c];
            prm[0] = gg.run_mean[c];
            prv[0] = gg.run_var[c];
            if (a.sn_two) {
                pw[2] = gf.w[2 * c];
                pw[3] = gf.w[2 * c + 1];
                pgam[1] = gf.gamma[c];
                pbet[1] = gf.beta[c];
                prm[1] = gf.run_mean[c];
                prv[1] = gf.run_var[c];
            }
        }

        // ---- load this wave's planes into registers (the only read of x)
        Raw<T, VEC> d[PPW][NV];
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            const int n = n0 + s;
            const T* pb = x + ((size_t)(n < N ? n : 0) * C + c) * ra.M;
            const int pbytes = n < N ? ra.M * (int)sizeof(T) : 0;  // past the batch end: every lane reads zeros
#pragma unroll
            for (int j = 0; j < NV; ++j) d[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(pb, pbytes, j0 + j), voff);
            if constexpr (EPI && !POST) {
                if (addend) {  // the op's input is x + addend, formed here and never written anywhere
                    const T* ab = addend + ((size_t)(n < N ? n : 0) * C + c) * ra.M;
#pragma unroll
                    for (int j = 0; j < NV; ++j)
                        d[s][j] = add_raw<T, VEC>(d[s][j], buf_load<T, VEC>(slot_rsrc<T, VEC>(ab, pbytes, j0 + j), voff));
                }
            }
        }
        Raw<T, VEC> ad[POST ? PPW : 1][POST ? NV : 1];  // POST addend planes
        auto fetch_addend = [&]() {
            if constexpr (POST) {
#pragma unroll
                for (int s = 0; s < PPW; ++s) {
                    const int n = n0 + s;
                    const T* ab = addend + ((size_t)(n < N ? n : 0) * C + c) * ra.M;
                    const int pbytes = n < N ? ra.M * (int)sizeof(T) : 0;
#pragma unroll
                    for (int j = 0; j < NV; ++j) ad[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(ab, pbytes, j0 + j), voff);
                }
            }
        };
        if constexpr (SOLO) fetch_addend();  // nothing to wait for: next to x

        // ---- exact two-pass statistics from registers; publish them to the cluster
        float solo_mean[SOLO ? PPW : 1], solo_m2[SOLO ? PPW : 1];
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            sg.forget();
            const int n = n0 + s;
            float pub[NG];
            if constexpr (!BOXED) {
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) sum += elem<T, VEC>(d[s][j], q);
                float tot[1] = {wave_sum(sum)};
                if constexpr (SPLIT) split_merge<1>(tot, xch, wave, lane);
                const float mean = tot[0] / (float)ra.M;
                float m2 = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (sg.valid(j)) {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            const float t = elem<T, VEC>(d[s][j], q) - mean;
                            m2 = fmaf(t, t, m2);
                        }
                    }
                float tq[1] = {wave_sum(m2)};
                if constexpr (SPLIT) split_merge<1>(tq, xch + 32, wave, lane);
                pub[0] = mean;
                pub[1] = tq[0];
            } else {
                float tot[1] = {wave_sum(slots_sum<T, VEC, NV>(d[s]))};  // invalid slots hold 0
                if constexpr (SPLIT) split_merge<1>(tot, xch, wave, lane);
                const float k = tot[0] / (float)a.M;
                float t[6];
                boxed_region_sums<T, VEC, NV>(d[s], sg, k, t);
#pragma unroll
                for (int m = 0; m < 6; ++m) t[m] = wave_sum(t[m]);
                if constexpr (SPLIT) split_merge<6>(t, xch + 32, wave, lane);
                boxed_moments(k, t, a.M, a.Mc, a.Ms, pub);
            }
            if constexpr (SOLO) {
                solo_mean[s] = pub[0];
                solo_m2[s] = pub[1];
                continue;
            }
            if (ra.epoch) {  // persistent context: lane m publishes pub[m] with the launch's tag
                if (n < N && lane < NG && (!SPLIT || wave == 0) && !(ra.fault && item == ra.K - 1)) {
                    float v = pub[0];
#pragma unroll
                    for (int m = 1; m < NG; ++m) v = (lane == m) ? pub[m] : v;
                    put_tagged(gran + ((size_t)c * N + n) * NG + lane, v, ra.epoch);
                }
            } else if (n < N && lane < NG / 2 && (!SPLIT || wave == 0) && !(ra.fault && item == ra.K - 1)) {  // lane m: (pub[2m], pub[2m+1])
                float lo = pub[0], hi = pub[1];
#pragma unroll
                for (int m = 1; m < NG / 2; ++m) {
                    lo = (lane == m) ? pub[2 * m] : lo;
                    hi = (lane == m) ? pub[2 * m + 1] : hi;
                }
                put_granule(gran + ((size_t)c * N + n) * (NG / 2) + lane, lo, hi);
            }
        }

        if constexpr (SOLO && !BOXED) {
            // every wave on its own: gate of each of its planes from the running statistics, apply, store
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                const int n = n0 + s;
                if (n >= N) continue;  // wave-uniform
                MomentsT<float> o;
                o.mu_c = o.mu_s = solo_mean[s];
                o.M2c = o.M2s = solo_m2[s];
                o.mu_o = o.M2o = 0.f;
                const FwdPlaneT<float> f = fwd_plane<float>(a, o, 0.f, 0.f);
                float g = 1.f, fg = 1.f;
                double zhg = 0.0, zhf = 0.0;
                if (a.sn_active) {
                    const double rg = (double)__builtin_amdgcn_rsqf(prv[0] + a.eps_bn);
                    zhg = ((double)pw[0] * (double)f.mu_p + (double)pw[1] * (double)f.sig_p - (double)prm[0]) * rg;
                    g = sigmoid_r<float>((float)((double)pgam[0] * zhg + (double)pbet[0]));
                    if (a.sn_two) {
                        const double rf = (double)__builtin_amdgcn_rsqf(prv[1] + a.eps_bn);
                        zhf = ((double)pw[2] * (double)f.mu_p + (double)pw[3] * (double)f.sig_p - (double)prm[1]) * rf;
                        fg = sigmoid_r<float>((float)((double)pgam[1] * zhf + (double)pbet[1]));
                    }
                    if (saved && n == 0 && lane == 0) {
                        saved[SV_ROWS * P + c] = rg;
                        saved[SV_ROWS * P + C + c] = (double)__builtin_amdgcn_rsqf(prv[1] + a.eps_bn);
                    }
                }
                const FwdCoefs cf = fwd_coefs<float>(a, f, g, fg);
                if (saved && lane == 0 && (!SPLIT || wave == 0)) {
                    const SvRec p = sv_rec(n, c, N);
                    store_fwd_plane<float>(saved, P, p, f, a.cn_active);
                    saved[sv_at(p, SV_G)] = g;
                    saved[sv_at(p, SV_ZH_G)] = zhg;
                    saved[sv_at(p, SV_F)] = fg;
                    saved[sv_at(p, SV_ZH_F)] = zhf;
                    if (a.save_coefs) store_fwd_coefs(saved, p, cf);
                }
                T* yb = y + ((size_t)n * C + c) * ra.M;
                const int pbytes = ra.M * (int)sizeof(T);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    float ov[VEC];
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        ov[q] = fmaf(cf.a_in, elem<T, VEC>(d[s][j], q) - cf.xr, cf.b_in);
                        if constexpr (POST) ov[q] += elem<T, VEC>(ad[s][j], q);
                        if constexpr (EPI) ov[q] = relu ? fmaxf(ov[q], 0.f) : ov[q];
                    }
                    buf_store<T, VEC>(slot_rsrc<T, VEC>(yb, pbytes, j0 + j), voff, pack<T, VEC>(ov));
                }
            }
            continue;
        }

        // ---- gather the whole channel's statistics
#if CNSN_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        CNSN_STAMP(1);
        if (threadIdx.x == 0) *gave_up = 0;
        __syncthreads();  // the previous item's readers of vals/zbuf are done
        CNSN_STAMP(2);
        unsigned passes_ = 0;
#if CNSN_SWEEP_SCALAR
        if (!sweep_granules_scalar(gran + (size_t)c * N * (NG / 2), N * NG, vals, ctl, ra.host_flag, ra.wait_ticks, wave, passes_))
            if (lane == 0) *gave_up = 1;
#else
        if (wave == 0) {
            const bool got = ra.epoch ? sweep_granules(gran + (size_t)c * N * NG, N * NG, vals, ctl, ra.host_flag, ra.wait_ticks,
                                                        passes_, ra.epoch, ra.ctl_idle)
                                      : sweep_granules(gran + (size_t)c * N * (NG / 2), N * (NG / 2), vals, ctl, ra.host_flag,
                                                        ra.wait_ticks, passes_, 0u, ra.ctl_idle);
            if (lane == 0 && !got) *gave_up = 1;
        }
#endif
        __syncthreads();
        if (*gave_up) {  // (workgroup-uniform) timed out: see sweep_granules; the planes still owed are marked with NaNs
#pragma unroll
            for (int s = 0; s < PPW; ++s)
                if (n0 + s < N && lane == 0 && (!SPLIT || wave == 0)) poison_plane<T, VEC>(y + ((size_t)(n0 + s) * C + c) * ra.M);
            return;
        }
#if CNSN_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        CNSN_STAMP(3);
        CNSN_NOTE(6, passes_);
        if constexpr (!SOLO) fetch_addend();  // the exchange is over: in flight during the algebra

        // per-plane algebra in float, cross-batch sums / normalisation in double (SPLIT: double throughout, see the backward)
        using R = typename std::conditional<SPLIT, double, float>::type;
        auto plane_of = [&](int n) {
            MomentsT<R> o;
            o.mu_c = vals[n * NG];
            o.M2c = vals[n * NG + 1];
            o.mu_o = BOXED ? vals[n * NG + 2] : 0.f;
            o.M2o = BOXED ? vals[n * NG + 3] : 0.f;
            o.mu_s = BOXED ? vals[n * NG + 4] : o.mu_c;
            o.M2s = BOXED ? vals[n * NG + 5] : o.M2c;
            R mu_sq = 0.f, M2_sq = 0.f;
            if (a.cn_active) {
                const int q = sperm[n];  // style source instance, same channel (cnsn.py:66,68)
                mu_sq = vals[q * NG + (BOXED ? 4 : 0)];
                M2_sq = vals[q * NG + (BOXED ? 5 : 1)];
            }
            return fwd_plane<R>(a, o, mu_sq, M2_sq);
        };

        // ---- SelfNorm gate statistics over the batch (every member computes them redundantly):
        //      one pass, sums taken about the pre-activation of instance 0
        double mg = 0, mf = 0, rg = 1, rf = 1, wg0 = 0, wg1 = 0, wf0 = 0, wf1 = 0;
        if (a.sn_active) {
            wg0 = pw[0];
            wg1 = pw[1];
            wf0 = pw[2];
            wf1 = pw[3];
            if (a.sn_training) {
#if CNSN_CHEAP_SHIFT
                // any value near the batch mean of z serves as the shift: plane 0's raw content moments
                const double sh_mu = (double)vals[0];
                const double sh_sd = (double)sqrt_r(div_r(vals[1], (float)(a.Mc - 1)) + a.eps_sn);
                const double zs_g = wg0 * sh_mu + wg1 * sh_sd, zs_f = wf0 * sh_mu + wf1 * sh_sd;
#else
                const FwdPlaneT<R> f0 = plane_of(0);
                const double zs_g = wg0 * (double)f0.mu_p + wg1 * (double)f0.sig_p;
                const double zs_f = wf0 * (double)f0.mu_p + wf1 * (double)f0.sig_p;
#endif
                double sz[4] = {0.0, 0.0, 0.0, 0.0};
                for (int n = threadIdx.x; n < N; n += kBlock) {
                    const FwdPlaneT<R> f = plane_of(n);
                    const double dg = wg0 * (double)f.mu_p + wg1 * (double)f.sig_p - zs_g;
                    const double df = wf0 * (double)f.mu_p + wf1 * (double)f.sig_p - zs_f;
                    sz[0] += dg;
                    sz[1] += dg * dg;
                    sz[2] += df;
                    sz[3] += df * df;
                }
                block_sum_d<4>(sz, red);
                mg = zs_g + sz[0] * a.inv_n;
                mf = zs_f + sz[2] * a.inv_n;
                double vg = (sz[1] - sz[0] * sz[0] * a.inv_n) * a.inv_n, vf = (sz[3] - sz[2] * sz[2] * a.inv_n) * a.inv_n;
                vg = vg > 0.0 ? vg : 0.0;
                vf = vf > 0.0 ? vf : 0.0;
                // rstd to float accuracy: a uniform scale on the normalised value (not amplified), and the
                // SAME number reaches the backward through `saved`
                rg = (double)__builtin_amdgcn_rsqf((float)(vg + (double)a.eps_bn));
                rf = (double)__builtin_amdgcn_rsqf((float)(vf + (double)a.eps_bn));
                if (k == 0 && threadIdx.x == 0) {
                    const double mom_ = a.momentum, unb = a.unbias_n;
                    gg.run_mean[c] = (float)((1.0 - mom_) * (double)prm[0] + mom_ * mg);
                    gg.run_var[c] = (float)((1.0 - mom_) * (double)prv[0] + mom_ * vg * unb);
                    if (a.sn_two) {
                        gf.run_mean[c] = (float)((1.0 - mom_) * (double)prm[1] + mom_ * mf);
                        gf.run_var[c] = (float)((1.0 - mom_) * (double)prv[1] + mom_ * vf * unb);
                    }
                    if (c == 0) {
                        bump_batches_tracked(gg.nbt);
                        if (a.sn_two) bump_batches_tracked(gf.nbt);
                    }
                }
            } else {
                mg = prm[0];
                rg = (double)__builtin_amdgcn_rsqf(prv[0] + a.eps_bn);
                if (a.sn_two) {
                    mf = prm[1];
                    rf = (double)__builtin_amdgcn_rsqf(prv[1] + a.eps_bn);
                }
            }
            if (saved && k == 0 && threadIdx.x == 0) {
                saved[SV_ROWS * P + c] = rg;
                saved[SV_ROWS * P + C + c] = rf;
            }
        }

        // ---- coefficients (and saved state) of the owned planes
#if CNSN_WAVE_COEF
        // lane s (< PPW) of every wave takes the wave's plane s; the apply loop reads them back with v_readlane
        FwdCoefs cfw{0.f, 0.f, 0.f, 0.f, 0.f};
        {
            const int n = n0 + (lane < PPW ? lane : 0);
            if (n < N) {
                const FwdPlaneT<R> f = plane_of(n);
                R g = 1.f, fg = 1.f;
                double zhg = 0.0, zhf = 0.0;
                if (a.sn_active) {
                    zhg = (wg0 * (double)f.mu_p + wg1 * (double)f.sig_p - mg) * rg;
                    g = sigmoid_r<R>((R)((double)pgam[0] * zhg + (double)pbet[0]));
                    if (a.sn_two) {
                        zhf = (wf0 * (double)f.mu_p + wf1 * (double)f.sig_p - mf) * rf;
                        fg = sigmoid_r<R>((R)((double)pgam[1] * zhf + (double)pbet[1]));
                    }
                }
                cfw = fwd_coefs<R>(a, f, g, fg);
                if (saved && lane < PPW) {
                    const SvRec p = sv_rec(n, c, N);
                    store_fwd_plane<R>(saved, P, p, f, a.cn_active);
                    saved[sv_at(p, SV_G)] = g;
                    saved[sv_at(p, SV_ZH_G)] = zhg;
                    saved[sv_at(p, SV_F)] = fg;
                    saved[sv_at(p, SV_ZH_F)] = zhf;
                    if (a.save_coefs) store_fwd_coefs(saved, p, cfw);
                }
            }
        }
#else
        if (threadIdx.x < OWN) {
            const int n = k * OWN + threadIdx.x;
            if (n < N) {
                const FwdPlaneT<R> f = plane_of(n);
                R g = 1.f, fg = 1.f;
                double zhg = 0.0, zhf = 0.0;
                if (a.sn_active) {
                    zhg = (wg0 * (double)f.mu_p + wg1 * (double)f.sig_p - mg) * rg;
                    g = sigmoid_r<R>((R)((double)pgam[0] * zhg + (double)pbet[0]));
                    if (a.sn_two) {
                        zhf = (wf0 * (double)f.mu_p + wf1 * (double)f.sig_p - mf) * rf;
                        fg = sigmoid_r<R>((R)((double)pgam[1] * zhf + (double)pbet[1]));
                    }
                }
                const FwdCoefs cf = fwd_coefs<R>(a, f, g, fg);
                float* o = ocoef + threadIdx.x * FC_ROWS;
                o[FC_A_IN] = cf.a_in;
                o[FC_XR] = cf.xr;
                o[FC_B_IN] = cf.b_in;
                o[FC_A_OUT] = cf.a_out;
                o[FC_B_OUT] = cf.b_out;
                if (saved) {
                    const SvRec p = sv_rec(n, c, N);
                    store_fwd_plane<R>(saved, P, p, f, a.cn_active);
                    saved[sv_at(p, SV_G)] = g;
                    saved[sv_at(p, SV_ZH_G)] = zhg;
                    saved[sv_at(p, SV_F)] = fg;
                    saved[sv_at(p, SV_ZH_F)] = zhf;
                    if (a.save_coefs) store_fwd_coefs(saved, p, cf);
                }
            }
        }
        __syncthreads();
#endif
        CNSN_STAMP(4);

        // ---- apply from registers, the only write of y
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            sg.forget();
            const int n = n0 + s;
            if (n < N) {
#if CNSN_WAVE_COEF
                const float a_in = lane_bcast(cfw.a_in, s), xr = lane_bcast(cfw.xr, s), b_in = lane_bcast(cfw.b_in, s),
                            a_out = lane_bcast(cfw.a_out, s), b_out = lane_bcast(cfw.b_out, s);
#else
                const float* o = ocoef + (SPLIT ? 0 : wave * PPW + s) * FC_ROWS;
                const float a_in = o[FC_A_IN], xr = o[FC_XR], b_in = o[FC_B_IN], a_out = o[FC_A_OUT],
                            b_out = o[FC_B_OUT];
#endif
                T* yb = y + ((size_t)n * C + c) * ra.M;
                const int pbytes = ra.M * (int)sizeof(T);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    float ov[VEC];
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float f = elem<T, VEC>(d[s][j], q);
                        if constexpr (!BOXED) {
                            ov[q] = fmaf(a_in, f - xr, b_in);
                        } else if ((q & 1) == 0) {  // both maps on the element PAIR, each element's picked by its mask word
                            const cnsn_f2_t f2 = {f, elem<T, VEC>(d[s][j], q + 1)};
                            const cnsn_f2_t r2 = pick_if2(sg.mask_c(j, q), sg.mask_c(j, q + 1), fma2(splat2(a_in), f2 - splat2(xr), splat2(b_in)),
                                                          fma2(splat2(a_out), f2, splat2(b_out)));
                            ov[q] = r2.x;
                            ov[q + 1] = r2.y;
                        }
                        if constexpr (POST) ov[q] += elem<T, VEC>(ad[s][j], q);
                        if constexpr (EPI) ov[q] = relu ? fmaxf(ov[q], 0.f) : ov[q];
                    }
                    buf_store<T, VEC>(slot_rsrc<T, VEC>(yb, pbytes, j0 + j), voff, pack<T, VEC>(ov));
                }
            }
        }
#if CNSN_PRIO
        __builtin_amdgcn_s_setprio(2);
#endif
        CNSN_STAMP(5);
    }
}

// ================================================================================================
// backward
// ================================================================================================
// POST (with EPI and ReLU, un-boxed): the mask is that of act(CNSN(x) + addend) — the addend is read next to G and x,
// used for the mask and dropped; the masked gradient is also the gradient of the addend and is written to d_addend.
template <typename T, int VEC, int NV, int PPW, bool BOXED, bool EPI = false, bool POST = false, bool SPLIT = false>
__global__ __launch_bounds__(kBlock, SPLIT ? 2 : POST ? 2 : EPI ? bwd_waves_epi((int)sizeof(T), VEC, NV) : bwd_waves(data_regs(sizeof(T), VEC, NV, PPW))) void resident_bwd_kernel(ResArgs ra, const T* __restrict__ gy,
                                                              const T* __restrict__ x, T* __restrict__ dx,
                                                              const int64_t* __restrict__ perm, GateDev gg, GateDev gf,
                                                              GateGradDev dgr, GateGradDev dfr,
                                                              unsigned long long* __restrict__ gran,
                                                              const double* __restrict__ saved,
                                                              unsigned* __restrict__ ctl,
                                                              const T* __restrict__ addend, int relu,
                                                              T* __restrict__ d_addend,
                                                              unsigned long long* __restrict__ clear, unsigned clear_n,
                                                              PermInline pin) {
    pipe_clear_other_region(clear, clear_n);
    constexpr int NS = BOXED ? 4 : 2;
    constexpr int OWN = SPLIT ? 1 : 4 * PPW;
    static_assert(!SPLIT || (PPW == 1 && !CNSN_WAVE_COEF), "a split plane is the workgroup's only plane");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MidArgs a = ra.mid;
    const int N = a.N, C = a.C;
    float* vals = (float*)smem;
    double* dtb = (double*)(smem + align16((size_t)N * NS * 4));
    int* iperm = (int*)((char*)dtb + align16((size_t)2 * N * 8));
    float* ocoef = (float*)((char*)iperm + align16((size_t)N * 4));
    double* red = (double*)((char*)ocoef + align16((size_t)OWN * BC_ROWS * 4));
    int* gave_up = (int*)(red + 4 * 4);
    float* xch = (float*)(gave_up + 4);                                // [2][4][8] (SPLIT)
    (void)xch;
    double* svd = red + 4 * 4 + 2 + 32;                                // [N][D_N]
    float* svf = (float*)((char*)svd + align16((size_t)N * D_N * 8));  // [N][F_N]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: plane bases stay in SGPRs
    const size_t P = (size_t)N * C;
    const int j0 = SPLIT ? wave * NV : 0;  // first slot of this wave within its plane
    const SlotGeom<VEC, NV, BOXED> sg(ra, lane + 64 * j0);
    constexpr int VB = VEC * (int)sizeof(T);  // bytes per vector
    const int voff = lane * VB;

    if (a.cn_active)  // plane r receives the style-statistic gradient of the plane that borrowed from it
        for (int n = threadIdx.x; n < N; n += kBlock) iperm[perm_at(perm, pin, n)] = n;
    startup_skew(ra);
#if CNSN_PRIO
    __builtin_amdgcn_s_setprio(2);
#endif

    int iter_ = -1;
    for (int item = cluster_block(ra.xcd); item < ra.items; item += gridDim.x) {
        const int c = item / ra.K, k = item - c * ra.K;
        const int n0 = SPLIT ? k : (k * 4 + wave) * PPW;
        ++iter_;
        CNSN_STAMP(0);

        // ---- everything the algebra will need from `saved` and the parameters is fetched FIRST, ahead of
        //      the bulk loads in this CU's memory queue, and staged in LDS / registers: a load issued in
        //      the algebra phase would wait behind every other workgroup's planes (several microseconds)
        float pw[4] = {0.f, 0.f, 0.f, 0.f}, pgam[2] = {0.f, 0.f};
        double prs[2] = {1.0, 1.0};
        if (a.sn_active) {
            pw[0] = gg.w[2 * c];
            pw[1] = gg.w[2 * c + 1];
            pgam[0] = gg.gamma[c];
            prs[0] = saved[SV_ROWS * P + c];
            if (a.sn_two) {
                pw[2] = gf.w[2 * c];
                pw[3] = gf.w[2 * c + 1];
                pgam[1] = gf.gamma[c];
                prs[1] = saved[SV_ROWS * P + C + c];
            }
        }
        float own_si[PPW], own_so[PPW];
        float own_fc[EPI ? PPW : 1][FC_ROWS];  // the forward's apply coefficients of the owned planes (ReLU mask)
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            const SvRec p = sv_rec((n0 + s < N ? n0 + s : 0), c, N);
            own_si[s] = (float)saved[sv_at(p, SV_MU_C)];
            own_so[s] = BOXED ? (float)saved[sv_at(p, SV_MU_O)] : 0.f;
            if constexpr (EPI) {
                if (relu) {
#pragma unroll
                    for (int r = 0; r < FC_ROWS; ++r) own_fc[s][r] = (float)saved[sv_at(p, SV_FC0 + r)];
                }
            }
        }
        for (int n = threadIdx.x; n < N; n += kBlock) {
            const SvRec p = sv_rec(n, c, N);
            float* sf = svf + n * F_N;
            double* sd = svd + n * D_N;
            sd[D_MU_C] = saved[sv_at(p, SV_MU_C)];
            sd[D_ZH_G] = saved[sv_at(p, SV_ZH_G)];
            sd[D_ZH_F] = saved[sv_at(p, SV_ZH_F)];
            const CnRowsT<float> cr = load_cn_rows<float>(a, saved, p, sd[D_MU_C]);
            sd[D_MU_S] = cr.mu_s;
            sf[F_A1] = cr.a1;
            sf[F_M_IN] = cr.m_in;
            sf[F_MU_O] = cr.mu_o;
            sf[F_MU_P] = (float)saved[sv_at(p, SV_MU_P)];
            sf[F_G] = (float)saved[sv_at(p, SV_G)];
            sf[F_F] = (float)saved[sv_at(p, SV_F)];
            sf[F_A] = cr.aa;
            sf[F_SIG_P] = (float)saved[sv_at(p, SV_SIG_P)];
            sf[F_SIG_C] = cr.sig_c;
            sf[F_M2C] = cr.M2c;
            sf[F_SIG_S] = cr.sig_s;
        }

        // ---- load G and x planes (the only reads)
        Raw<T, VEC> dg_[PPW][NV], dx_[PPW][NV];
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            sg.forget();
            const int n = n0 + s;
            const size_t off = ((size_t)(n < N ? n : 0) * C + c) * ra.M;
            const int pbytes = n < N ? ra.M * (int)sizeof(T) : 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                dg_[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(gy + off, pbytes, j0 + j), voff);
                dx_[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(x + off, pbytes, j0 + j), voff);
            }
            if constexpr (EPI) {
                if constexpr (!POST) {
                    if (addend) {
#pragma unroll
                        for (int j = 0; j < NV; ++j)
                            dx_[s][j] = add_raw<T, VEC>(dx_[s][j],
                                                        buf_load<T, VEC>(slot_rsrc<T, VEC>(addend + off, pbytes, j0 + j), voff));
                    }
                }
                Raw<T, VEC> ad[POST ? NV : 1];
                if constexpr (POST) {
#pragma unroll
                    for (int j = 0; j < NV; ++j) ad[j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(addend + off, pbytes, j0 + j), voff);
                }
                if (relu) {  // shut the gradient where the forward's output was not positive: the forward affine
                             // is re-evaluated with the coefficients the forward itself used
                    const float a_in = own_fc[s][FC_A_IN], xr = own_fc[s][FC_XR], b_in = own_fc[s][FC_B_IN],
                                a_out = own_fc[s][FC_A_OUT], b_out = own_fc[s][FC_B_OUT];
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        float gm[VEC];
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            const float X = elem<T, VEC>(dx_[s][j], q);
                            float t = fmaf(a_in, X - xr, b_in);
                            if constexpr (BOXED) t = pick_if(sg.mask_c(j, q), t, fmaf(a_out, X, b_out));
                            if constexpr (POST) t += elem<T, VEC>(ad[j], q);
                            gm[q] = relu_open_r<T>(t) ? elem<T, VEC>(dg_[s][j], q) : 0.f;
                        }
                        dg_[s][j] = pack<T, VEC>(gm);
                        if constexpr (POST)  // the masked gradient is the addend's gradient
                            buf_store<T, VEC>(slot_rsrc<T, VEC>(d_addend + off, pbytes, j0 + j), voff, dg_[s][j]);
                    }
                }
            }
        }

        // ---- per-plane sums of G against x (shifted by the saved means, as pass A' does); publish
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            sg.forget();
            const int n = n0 + s;
            const float si = own_si[s], so = own_so[s];
            float acc[NS];
#pragma unroll
            for (int m = 0; m < NS; ++m) acc[m] = 0.f;
            if constexpr (BOXED) {  // whole-plane sums in acc[2..3], content-box sums in acc[0..1], both about float(mu_c)
                boxed_bwd_sums<T, VEC, NV>(dg_[s], dx_[s], [&](int j) { return sg.valid(j); }, [&](int j, int q) { return sg.mask_c(j, q); }, si, acc);
            } else
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (sg.valid(j)) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float G = elem<T, VEC>(dg_[s][j], q), X = elem<T, VEC>(dx_[s][j], q);
                        acc[0] += G;
                        acc[1] = fmaf(G, X - si, acc[1]);
                    }
                }
#pragma unroll
            for (int m = 0; m < NS; ++m) acc[m] = wave_sum(acc[m]);
            if constexpr (BOXED) {  // outside the box = whole plane - box, re-centred on float(mu_o):
                                    // sum_o G*(X - so) = sum_o G*(X - si) + (si - so) * sum_o G
                const float o1 = acc[2] - acc[0];
                acc[3] = (acc[3] - acc[1]) + (si - so) * o1;
                acc[2] = o1;
            }
            if constexpr (SPLIT) split_merge<NS>(acc, xch, wave, lane);
            if (ra.epoch) {
                if (n < N && lane < NS && (!SPLIT || wave == 0) && !(ra.fault && item == ra.K - 1)) {
                    float v = acc[0];
#pragma unroll
                    for (int m = 1; m < NS; ++m) v = (lane == m) ? acc[m] : v;
                    put_tagged(gran + ((size_t)c * N + n) * NS + lane, v, ra.epoch);
                }
            } else if (n < N && lane < NS / 2 && (!SPLIT || wave == 0) && !(ra.fault && item == ra.K - 1)) {
                float lo = acc[0], hi = acc[1];
#pragma unroll
                for (int m = 1; m < NS / 2; ++m) {
                    lo = (lane == m) ? acc[2 * m] : lo;
                    hi = (lane == m) ? acc[2 * m + 1] : hi;
                }
                put_granule(gran + ((size_t)c * N + n) * (NS / 2) + lane, lo, hi);
            }
        }

#if CNSN_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        CNSN_STAMP(1);
        if (threadIdx.x == 0) *gave_up = 0;
        __syncthreads();
        CNSN_STAMP(2);
        unsigned passes_ = 0;
#if CNSN_SWEEP_SCALAR
        if (!sweep_granules_scalar(gran + (size_t)c * N * (NS / 2), N * NS, vals, ctl, ra.host_flag, ra.wait_ticks, wave, passes_))
            if (lane == 0) *gave_up = 1;
#else
        if (wave == 0) {
            const bool got = ra.epoch ? sweep_granules(gran + (size_t)c * N * NS, N * NS, vals, ctl, ra.host_flag, ra.wait_ticks,
                                                        passes_, ra.epoch, ra.ctl_idle)
                                      : sweep_granules(gran + (size_t)c * N * (NS / 2), N * (NS / 2), vals, ctl, ra.host_flag,
                                                        ra.wait_ticks, passes_, 0u, ra.ctl_idle);
            if (lane == 0 && !got) *gave_up = 1;
        }
#endif
        __syncthreads();
        if (*gave_up) {  // (workgroup-uniform) timed out: see sweep_granules; the planes still owed are marked with NaNs
#pragma unroll
            for (int s = 0; s < PPW; ++s)
                if (n0 + s < N && lane == 0 && (!SPLIT || wave == 0)) poison_plane<T, VEC>(dx + ((size_t)(n0 + s) * C + c) * ra.M);
            return;
        }
#if CNSN_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        CNSN_STAMP(3);
        CNSN_NOTE(6, passes_);

        // per-plane algebra in float (batch sums and the dz line in double); SPLIT planes are 4-16x larger, their sums
        // carry 4-16x the absolute rounding into the BatchNorm-backward cancellation, and one workgroup has ONE plane's
        // algebra to do: double there (tests/test_gpu_parity.py, (2,4,128,96): grad of g_fc.weight 1.4e-2 -> within 2x of the fp32 oracle)
        using R = typename std::conditional<SPLIT, double, float>::type;
        auto sums_of = [&](int n) {
            return fix_sums<R>(a, vals[n * NS], vals[n * NS + 1], BOXED ? vals[n * NS + 2] : 0.f,
                               BOXED ? vals[n * NS + 3] : 0.f, svd[n * D_N + D_MU_C], (double)svf[n * F_N + F_MU_O]);
        };

        // ---- gate backward: dt for every instance of the channel, batch sums (all members)
        double s4[4] = {0, 0, 0, 0};
        BnBwd b{};
        if (a.sn_active) {
            for (int n = threadIdx.x; n < N; n += kBlock) {
                const float* sf = svf + n * F_N;
                R dtg, dtf;
                gate_dt<R>(a, sums_of(n), sf[F_A1], sf[F_M_IN], sf[F_MU_O], sf[F_MU_P], sf[F_G], sf[F_F], dtg, dtf);
                s4[0] += (double)dtg;
                s4[1] += (double)dtg * svd[n * D_N + D_ZH_G];
                s4[2] += (double)dtf;
                s4[3] += (double)dtf * svd[n * D_N + D_ZH_F];
                dtb[n] = dtg;
                dtb[N + n] = dtf;
            }
            block_sum_d<4>(s4, red);
            b.s_dt_g = s4[0];
            b.s_dtz_g = s4[1];
            b.s_dt_f = s4[2];
            b.s_dtz_f = s4[3];
            b.wg0 = pw[0];
            b.wg1 = pw[1];
            b.kg = (double)pgam[0] * prs[0];
            b.wf0 = pw[2];
            b.wf1 = pw[3];
            b.kf = (double)pgam[1] * prs[1];
        }

        auto bwd_of = [&](int n) {
            const float* sf = svf + n * F_N;
            return bwd_plane<R>(a, b, sums_of(n), a.sn_active ? dtb[n] : 0.0, a.sn_active ? dtb[N + n] : 0.0,
                                svd[n * D_N + D_ZH_G], svd[n * D_N + D_ZH_F], sf[F_G], sf[F_F], sf[F_A], sf[F_A1],
                                sf[F_M_IN], sf[F_MU_P], sf[F_SIG_P], sf[F_SIG_C], sf[F_M2C]);
        };

        // ---- parameter gradients of the channel: one member per channel (rotating) does the sums
        if (a.sn_active && k == c % ra.K) {
            double sw[4] = {0, 0, 0, 0};
            for (int n = threadIdx.x; n < N; n += kBlock) {
                const BwdPlaneT<R> o = bwd_of(n);
                const double mu_p = svf[n * F_N + F_MU_P], sig_p = svf[n * F_N + F_SIG_P];
                sw[0] += (double)o.dz_g * mu_p;
                sw[1] += (double)o.dz_g * sig_p;
                sw[2] += (double)o.dz_f * mu_p;
                sw[3] += (double)o.dz_f * sig_p;
            }
            block_sum_d<4>(sw, red);
            if (threadIdx.x == 0) {
                dgr.dgamma[c] = (float)s4[1];
                dgr.dbeta[c] = (float)s4[0];
                dgr.dw[2 * c] = (float)sw[0];
                dgr.dw[2 * c + 1] = (float)sw[1];
                if (a.sn_two) {
                    dfr.dgamma[c] = (float)s4[3];
                    dfr.dbeta[c] = (float)s4[2];
                    dfr.dw[2 * c] = (float)sw[2];
                    dfr.dw[2 * c + 1] = (float)sw[3];
                }
            }
        }

        // ---- coefficients of dx for the owned planes
#if CNSN_WAVE_COEF
        BwdCoefs cfw{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        {
            const int n = n0 + (lane < PPW ? lane : 0);  // lane s (< PPW) takes the wave's plane s
            if (n < N) {
                const float* sf = svf + n * F_N;
                const BwdPlaneT<R> o = bwd_of(n);
                R Emu = 0.f, Esig = 0.f;
                if (a.cn_active) {
                    const BwdPlaneT<R> src = bwd_of(iperm[n]);  // the plane that used (n,c) as its style
                    Emu = src.Emu;
                    Esig = src.Esig;
                }
                cfw = bwd_coefs<R>(a, o, Emu, Esig, sf[F_G], sf[F_A1], sf[F_M_IN], sf[F_MU_P], svd[n * D_N + D_MU_C],
                                   sf[F_SIG_C], svd[n * D_N + D_MU_S], sf[F_SIG_S]);
            }
        }
        __syncthreads();  // the next item's prefetch overwrites svd / svf: every wave must be done reading them
#else
        if (threadIdx.x < OWN) {
            const int n = k * OWN + threadIdx.x;
            if (n < N) {
                const float* sf = svf + n * F_N;
                const BwdPlaneT<R> o = bwd_of(n);
                R Emu = 0.f, Esig = 0.f;
                if (a.cn_active) {
                    const BwdPlaneT<R> src = bwd_of(iperm[n]);  // the plane that used (n,c) as its style
                    Emu = src.Emu;
                    Esig = src.Esig;
                }
                const BwdCoefs cf =
                    bwd_coefs<R>(a, o, Emu, Esig, sf[F_G], sf[F_A1], sf[F_M_IN], sf[F_MU_P], svd[n * D_N + D_MU_C],
                                 sf[F_SIG_C], svd[n * D_N + D_MU_S], sf[F_SIG_S]);
                float* oc = ocoef + threadIdx.x * BC_ROWS;
                oc[BC_CG_IN] = cf.cG_in;
                oc[BC_CX_IN] = cf.cX_in;
                oc[BC_XR_IN] = cf.xr_in;
                oc[BC_C0_IN] = cf.c0_in;
                oc[BC_CG_OUT] = cf.cG_out;
                oc[BC_CX_OUT] = cf.cX_out;
                oc[BC_XR_OUT] = cf.xr_out;
                oc[BC_C0_OUT] = cf.c0_out;
                oc[BC_ES] = cf.eS;
                oc[BC_XS] = cf.xs;
                oc[BC_E0] = cf.e0;
            }
        }
        __syncthreads();
#endif
        CNSN_STAMP(4);

        // ---- apply from registers, the only write of dx
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            sg.forget();
            const int n = n0 + s;
            if (n < N) {
#if CNSN_WAVE_COEF
                const float cG_i = lane_bcast(cfw.cG_in, s), cX_i = lane_bcast(cfw.cX_in, s), xr_i = lane_bcast(cfw.xr_in, s),
                            c0_i = lane_bcast(cfw.c0_in, s);
                float cG_o = 0.f, cX_o = 0.f, xr_o = 0.f, c0_o = 0.f, eS = 0.f, xs = 0.f, e0 = 0.f;
                if constexpr (BOXED) {
                    cG_o = lane_bcast(cfw.cG_out, s), cX_o = lane_bcast(cfw.cX_out, s), xr_o = lane_bcast(cfw.xr_out, s),
                    c0_o = lane_bcast(cfw.c0_out, s), eS = lane_bcast(cfw.eS, s), xs = lane_bcast(cfw.xs, s),
                    e0 = lane_bcast(cfw.e0, s);
                }
#else
                const float* oc = ocoef + (SPLIT ? 0 : wave * PPW + s) * BC_ROWS;
                const float cG_i = oc[BC_CG_IN], cX_i = oc[BC_CX_IN], xr_i = oc[BC_XR_IN], c0_i = oc[BC_C0_IN];
                const float cG_o = oc[BC_CG_OUT], cX_o = oc[BC_CX_OUT], xr_o = oc[BC_XR_OUT], c0_o = oc[BC_C0_OUT];
                const float eS = oc[BC_ES], xs = oc[BC_XS], e0 = oc[BC_E0];
#endif
                T* db = dx + ((size_t)n * C + c) * ra.M;
                const int pbytes = ra.M * (int)sizeof(T);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    float ov[VEC];
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float G = elem<T, VEC>(dg_[s][j], q), X = elem<T, VEC>(dx_[s][j], q);
                        float v;
                        if constexpr (!BOXED) {
                            v = fmaf(cG_i, G, fmaf(cX_i, X - xr_i, c0_i));
                        } else {
                            if ((q & 1) == 0) {  // the three affine maps on the element PAIR, then each element's by its mask words
                                const cnsn_f2_t G2 = {G, elem<T, VEC>(dg_[s][j], q + 1)}, X2 = {X, elem<T, VEC>(dx_[s][j], q + 1)};
                                const cnsn_f2_t vi = fma2(splat2(cG_i), G2, fma2(splat2(cX_i), X2 - splat2(xr_i), splat2(c0_i)));
                                const cnsn_f2_t vo = fma2(splat2(cG_o), G2, fma2(splat2(cX_o), X2 - splat2(xr_o), splat2(c0_o)));
                                const cnsn_f2_t r2 = pick_if2(sg.mask_c(j, q), sg.mask_c(j, q + 1), vi, vo) +
                                                     keep_if2(fma2(splat2(eS), X2 - splat2(xs), splat2(e0)), sg.mask_s(j, q), sg.mask_s(j, q + 1));
                                ov[q] = r2.x;
                                ov[q + 1] = r2.y;
                            }
                            continue;
                        }
                        ov[q] = v;
                    }
                    buf_store<T, VEC>(slot_rsrc<T, VEC>(db, pbytes, j0 + j), voff, pack<T, VEC>(ov));
                }
            }
        }
#if CNSN_PRIO
        __builtin_amdgcn_s_setprio(2);
#endif
        CNSN_STAMP(5);
    }
}

// ------------------------------------------------------------------------------------------------
// host entry points (defined in cnsn_resident.hip)
// ------------------------------------------------------------------------------------------------
struct ResPlan {
    bool ok;
    int vec, nv, ppw, K;
};
// can the resident strategy run this problem?  (auto = apply the profitability heuristics too)
ResPlan resident_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, bool backward);
// pipelined forward (cnsn_resident_pipe.hip): the next item's loads are in flight across the exchange of the current one;
// ok only where resident_plan(forward) is ok too.  *npark: slots of an item parked in LDS.
ResPlan resident_pipe_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, int* npark);
int resident_pipe_forward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* x,
                          const int64_t* perm, GateDev g, GateDev f, void* y, double* saved, void* workspace,
                          hipStream_t stream);
ResPlan resident_pipe_bwd_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, int* npark);
int resident_pipe_backward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* gy,
                           const void* x, const int64_t* perm, GateDev g, GateDev f, const double* saved, void* dx,
                           GateGradDev dg, GateGradDev df, void* workspace, hipStream_t stream);
// planes of 1025..4096 vectors: one plane per workgroup, split over its four waves (cnsn_resident_split.hip);
// add: ADD_NONE or ADD_POST (un-boxed), with or without ReLU
ResPlan resident_split_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, int add, int relu, bool backward);
int resident_split_forward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, int add, int relu,
                           const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f, void* y,
                           double* saved, void* workspace, hipStream_t stream);
int resident_split_backward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, int add, int relu,
                            const void* gy, const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f,
                            const double* saved, void* dx, void* d_addend, GateGradDev dg, GateGradDev df, void* workspace,
                            hipStream_t stream);

// both return CNSN_OK, a hipError_t, or CNSN_E_UNSUPPORTED (caller falls back to two-pass)
int resident_forward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* x,
                     const int64_t* perm, GateDev g, GateDev f, void* y, double* saved, void* workspace,
                     hipStream_t stream);
int resident_backward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* gy,
                      const void* x, const int64_t* perm, GateDev g, GateDev f, const double* saved, void* dx,
                      GateGradDev dg, GateGradDev df, void* workspace, hipStream_t stream);
size_t resident_workspace_bytes(const cnsn_problem_t& p, bool boxed);

// A persistent cluster grid assumes that ALL its workgroups become resident.  Two such grids started on DIFFERENT
// streams of one device could each end up partially resident and wait for each other (until the bounded spin
// traps).  Cluster launches of a process are therefore chained per device: a launch on another stream than the
// previous one first waits (stream-side, hipStreamWaitEvent) for the previous one's completion event.  Same-stream
// sequences — the normal case — pay one hipEventRecord.  Streams under graph capture are left alone (a captured
// graph replays on one stream).  This is the only state the library keeps.
// Host-visible word (pinned memory, allocated once per process on the first cluster launch) that the kernels bump
// when a bounded wait ran out; NULL when it could not be allocated (e.g. first use inside a stream capture).
// Fill the launch-argument form of the batch permutation.  perm (device) given: off.  Else from p.perm_host (N <= 1024,
// entries checked).  Returns CNSN_OK, CNSN_E_NULL (CrossNorm armed and neither form given) or CNSN_E_SHAPE (an index out of
// range).  `out` is a caller-provided object (2 KB: the launchers keep one per thread).
inline int perm_inline_fill(const cnsn_problem_t& p, const int64_t* perm, PermInline* out) {
    out->on = 0;
    if (!p.cn_active || perm) return CNSN_OK;
    if (!p.perm_host || p.N > kPermInlineMax) return CNSN_E_NULL;
    for (int n = 0; n < p.N; ++n) {
        const int64_t q = p.perm_host[n];
        if (q < 0 || q >= p.N) return CNSN_E_SHAPE;
        out->v[n] = (unsigned short)q;
    }
    out->on = 1;
    return CNSN_OK;
}
inline PermInline* perm_inline_scratch() {
    static thread_local PermInline pin{};  // (zero-initialised once; only v[0..N) is ever rewritten)
    return &pin;
}

unsigned* resident_host_flag();
// launches that timed out so far (0: none); after the first one AUTO stops choosing this strategy
int resident_timeouts();
bool resident_degraded();  // a launch gave up since the last cnsn_resident_rearm (or ever)
int resident_rearm();      // forgive the time-outs so far; returns how often this process has re-armed
void resident_set_wait_ms(int ms);   // cnsn_set_wait_ms
void resident_set_headroom_cus(int n);  // cnsn_set_headroom_cus
int resident_headroom_cus();            // compute units the persistent grids leave free (CNSN_HEADROOM_CUS wins over the setter)
long long resident_wait_ticks();     // bound of a cluster wait in 100 MHz ticks
// AUTO may choose the cluster kernels: not switched off (cnsn_resident_enable(0) / CNSN_RESIDENT=0), no time-out seen
bool resident_auto_enabled();
void resident_set_enabled(bool on);

// Exchange area of a launch: the caller's persistent context when it is usable (big enough, stream not capturing) — then
// `epoch` is the context's next launch number (> 0) and nothing is cleared — or the workspace with epoch 0 (memset).
// A wrap of the counter clears the context on the stream first.  Returns the area's base.
struct ExchangeArea {
    void* base;
    unsigned epoch;
};
// prefer_context: use the context whatever the tensor size (kernels whose gather is a few hundred bytes per item)
ExchangeArea resident_exchange_area(const cnsn_problem_t& p, size_t tagged_bytes, void* workspace, hipStream_t stream,
                                    bool prefer_context = false);
void resident_context_forget(void* context);

// Untagged granules WITHOUT a fill launch (pipelined kernels, tensors of 64 MiB and more — where a tagged context costs more
// than it saves): two regions of kPongRegion bytes at the END of the persistent context, used alternately.  A launch
// exchanges through one and its workgroups clear the other (same extent) for the launch after it; the host keeps, per
// context, which region comes next and how many bytes of each are known to be clean, and says when a fill is needed after
// all (first use, a larger extent than the previous launch cleared, a re-initialised context).  Tagged launches never
// reach into the regions (resident_exchange_area leaves them out of what it offers).  Call inside the ResidentChain.
constexpr size_t kPongRegion = (size_t)2 << 20;
struct PongArea {
    void* base;                  // region of this launch: control block + granules
    unsigned long long* clear;   // the other region
    unsigned clear_qwords;
    bool need_fill;              // `base` is not known to be clean over fill_bytes: memset it first
};
bool resident_pong_acquire(const cnsn_problem_t& p, size_t fill_bytes, hipStream_t stream, PongArea* out);
void resident_pong_commit(const cnsn_problem_t& p, size_t fill_bytes);  // the launch was issued: the other region will be clean

// The grid barrier of the single-launch channels-last kernels (cnsn_nhwc_fused_kernels.h): a BARRIER BLOCK of kBarBlock bytes,
// 17 counters in cache lines of their own — eight group counters (a workgroup arrives at counter blockIdx % 8: the XCD it runs
// on as far as anybody has observed, which matters for speed only), one top counter the last arriver of each group bumps, eight
// generation words the last arriver at the top writes and the members of a group poll.  One word with 1 024 arrivers and
// 1 024 pollers serialises at ~12 ns per access; sharded like this a barrier of 1 024 workgroups costs ~10 us
// (MI355X_MICROARCH.md, rows barrier-counter / barrier-xcd).  Every counter only GROWS: in a persistent context the host
// keeps, per context, what the launches so far have left (`group_base` arrivals per group counter, `bar_base` barriers) and
// nothing is cleared between launches; the block lies in front of the two granule regions at the end of the context
// (cnsn_context_bytes counts it, resident_exchange_area keeps tagged granules out of it).  Without a usable context (none,
// CNSN_CONTEXT=0, stream capture, too small) the block is `workspace_bar` and the caller zeroes it in front of the launch
// (`need_fill`).  The grid must be a multiple of 8 (equal groups).  Control word (`ctl`, the time-out flag every waiting
// workgroup of the library watches) idle value: 0 either way.  Call inside the ResidentChain.
constexpr size_t kBarBlock = 4096;
constexpr int kBarLine = 128;   // bytes between two counters
constexpr int kBarTop = 8;      // line of the top counter; lines 0..7 group counters, 9..16 generation words, 17: control word
constexpr int kBarGen = 9;
constexpr int kBarCtl = 17;     // (workspace form only: a context's control word is its word 0)
struct BarArea {
    unsigned* ctl;
    char* block;
    unsigned long long group_base;  // arrivals every group counter holds before this launch
    unsigned long long bar_base;    // barriers before this launch (top counter = 8 x, generation words = 1 x)
    bool need_fill;
};
BarArea resident_bar_area(const cnsn_problem_t& p, void* workspace_bar, hipStream_t stream, int grid, int barriers);

struct ResidentChain {
    explicit ResidentChain(hipStream_t stream);
    ~ResidentChain();
    hipStream_t stream_;
    int dev_;
    bool active_;
};

}  // namespace cnsn
