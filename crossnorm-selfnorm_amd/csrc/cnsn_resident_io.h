// Register discipline shared by the pipelined and the partial-moment cluster kernels (moved out of
// cnsn_resident_sn_kernels.h in round 5, when the pipelined forward adopted it): ONE buffer descriptor per tensor, kernel
// arguments re-read where they are used, wave-uniform values the optimiser may not hoist, phase fences for the register
// allocator.
#pragma once
#include "cnsn_resident_kernels.h"

namespace cnsn {

// ---- plane access: ONE buffer descriptor per tensor ----------------------------------------------------------------------
// The general resident kernels build a descriptor per plane and slot (cnsn_resident_kernels.h: 4 SGPRs and 64-bit scalar
// arithmetic each); with many planes per wave in flight the compiler then spills SGPRs into VGPR lanes, and those
// v_writelane / v_readlane instructions are VALU issue slots — 35-40 % of the vector instructions of the 28x28 and 14x14
// kernels of this family (ISA count), on classes whose per-plane fixed cost already sits near the VALU budget of a
// bandwidth-bound kernel.  Here the descriptor covers the whole (N, C, H, W) tensor and a plane is reached through the
// instruction's scalar offset (one SGPR, 32-bit arithmetic).  gfx950 range-checks soffset + voffset against num_records
// (probed: tools/soffset_probe.hip), so:
//   * a plane past the batch end, or an empty slot, gets soffset = tensor bytes: loads return zeros, stores are dropped;
//   * lanes past the end of the plane in its partly filled slot get voffset = tensor bytes (same effect).  Slots are
//     RIGHT-ALIGNED — slot j holds vectors (j - shift)*64 + lane with shift = NV - slots needed — so that the partly filled
//     slot is always slot NV-1 and the choice between the two voffset registers is made at compile time;
//   * everything stays below 2^32 as long as the tensor is smaller than 1 GiB (the plans check): the largest sum an
//     instruction can see is dead + dead (a dead plane AND a dead lane) + s*stride + a slot offset = 2*bytes + less than
//     bytes + a few KB; with bytes < 2^30 that is < 2^32 - no wrap back into the tensor (round-5 advisor: below 2 GiB it could).
template <typename T, int VEC, int NV>
struct PlaneIo {
    static constexpr int VB = VEC * (int)sizeof(T), SLOT = 64 * VB;
    int voff_full, voff_part;  // per lane
    int shift, tail;           // wave-uniform: empty leading slots; valid lanes of the last slot (1..64)
    unsigned dead;             // tensor bytes
    int lane;
    __device__ __forceinline__ PlaneIo(const ResArgs& ra, int N, int C, int lane_) : lane(lane_) {
        const int need = (ra.nvec + 63) >> 6;
        shift = NV - need;
        tail = ra.nvec - (need - 1) * 64;
        dead = (unsigned)N * (unsigned)C * (unsigned)ra.M * (unsigned)sizeof(T);
        voff_full = lane * VB;
        voff_part = lane < tail ? lane * VB : (int)dead;
    }
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t tensor(const T* base) const {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)dead, 0x00020000);
    }
    // byte offset of plane (n, c); `live` false: nothing there (reads zeros, drops stores)
    __device__ __forceinline__ unsigned plane(int n, int c, int C, int M, bool live) const {
        return live ? ((unsigned)n * (unsigned)C + (unsigned)c) * (unsigned)M * (unsigned)sizeof(T) : dead;
    }
    // the planes (n0 + s, c), s = 0.. of one wave: `span` = offset of the first (or `dead` when the item does not exist),
    // consecutive ones `stride` = C*M*sizeof(T) apart, the first `nlive` of them inside the batch.  Two scalar instructions
    // per plane and nothing to keep: the per-plane conditions and offsets of the unrolled loops are NOT worth an SGPR each
    // (the compiler kept them all and spilled)
    __device__ __forceinline__ unsigned span(int n0, int c, int C, int M, bool exists) const {
        return exists ? ((unsigned)n0 * (unsigned)C + (unsigned)c) * (unsigned)M * (unsigned)sizeof(T) : dead;
    }
    __device__ __forceinline__ unsigned at(unsigned span_off, unsigned stride, int s, int nlive) const {
        return (s < nlive ? span_off : dead) + (unsigned)s * stride;  // (dead + s*stride < 2^31: the tensor is below 1 GiB)
    }
    __device__ __forceinline__ bool valid(int j) const { return j >= shift && (j < NV - 1 || lane < tail); }
    __device__ __forceinline__ int voff(int j) const { return j == NV - 1 ? voff_part : voff_full; }
    __device__ __forceinline__ int soff(unsigned plane_off, int j) const {
        return (int)(j >= shift ? plane_off + (unsigned)((j - shift) * SLOT) : dead);
    }
    __device__ __forceinline__ Raw<T, VEC> load(__amdgpu_buffer_rsrc_t r, unsigned plane_off, int j) const {
        if constexpr (VB == 16)
            return __builtin_amdgcn_raw_buffer_load_b128(r, voff(j), soff(plane_off, j), CNSN_RES_LOAD_AUX);
        else
            return __builtin_amdgcn_raw_buffer_load_b64(r, voff(j), soff(plane_off, j), CNSN_RES_LOAD_AUX);
    }
    // Stores carry the plane offset in the VECTOR offset (one v_add_u32), not in soffset: a 16-byte buffer store with an SGPR
    // soffset may have its data registers overwritten by the next VALU instruction before it has read them — hipcc pads that
    // hazard only for stores WITHOUT a register soffset (GCNHazardRecognizer assumes the other form is safe), and on gfx950
    // it is not: the first two elements of lanes 12-15 of every row of 16 came out as the next slot's numbers, now and then
    // (tests/test_gpu_sn_cluster.py::test_full_pipeline_many_channels, run-to-run differences).
    __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, unsigned plane_off, int j, const Raw<T, VEC>& v) const {
        const int vo = voff(j) + soff(plane_off, j);  // (dead + dead < 2^32)
        if constexpr (VB == 16)
            __builtin_amdgcn_raw_buffer_store_b128(v, r, vo, 0, CNSN_RES_STORE_AUX);
        else
            __builtin_amdgcn_raw_buffer_store_b64(v, r, vo, 0, CNSN_RES_STORE_AUX);
    }
};

// ---- kernel arguments are read where they are used --------------------------------------------------------------------------
// These kernels have more wave-uniform state than the 102 SGPRs of a wave: ~46 dwords of ResArgs, a dozen pointers, the
// tensor descriptors, the item bookkeeping, the per-plane conditions of the unrolled loops.  Left to itself the compiler
// keeps every kernel argument in an SGPR from the first instruction to the last and spills the excess into VGPR lanes:
// 35-40 % of the VALU instructions of the hot loops were v_readlane / v_writelane of spilled scalars (ISA count of the 28x28
// and 14x14 classes, whose per-plane fixed cost already sits at the VALU budget of a bandwidth-bound kernel).  So the
// arguments travel as ONE struct, and the code re-reads a field from the kernarg segment (s_load through the scalar cache)
// at the place that needs it: `kargs_now` hides the pointer behind an empty asm, so the loads can neither be hoisted to the
// top of the kernel nor merged with earlier ones, and the live ranges stay inside one phase of one iteration.
template <typename KA>
__device__ __forceinline__ const KA* kargs_now() {
    typedef const __attribute__((address_space(4))) KA* KP;
    KP p = (KP)__builtin_amdgcn_kernarg_segment_ptr();
#ifndef SNX_NO_LAUNDER
    asm volatile("" : "+s"(p));
#endif
    return (const KA*)p;
}

// a wave-uniform value the optimiser may not look through: what is computed from it inside a loop stays inside the loop.
// (Loop-invariant scalars of the unrolled plane loops — "plane s is inside the batch", s * stride — were hoisted out of the
// item loop, one SGPR or SGPR pair per plane, and spilled; recomputing them costs one scalar instruction each.)
__device__ __forceinline__ int opaque_s(int v) {
    v = __builtin_amdgcn_readfirstlane(v);
#ifndef SNX_NO_OPAQUE
    asm volatile("" : "+s"(v));
#endif
    return v;
}
__device__ __forceinline__ unsigned opaque_s(unsigned v) { return (unsigned)opaque_s((int)v); }

// A phase boundary for the register allocator: every SGPR from s16 up is declared clobbered, so whatever lives across this
// point is parked in a VGPR lane ONCE and comes back right before its next use — which for most of it is not in the phase
// that follows.  Without it the allocator hands the registers to the long-lived values first (loop invariants it hoisted,
// the item bookkeeping, the other phases' operands) and the plane loops, which touch a score of SGPRs, reload their
// descriptors and offsets from spill lanes at every use (v_readlane is a VALU slot: 13-15 of them per plane, ISA count).
#ifndef SNX_PHASE_FENCE
#define SNX_PHASE_FENCE 1
#endif
__device__ __forceinline__ void snx_phase_fence() {
#if SNX_PHASE_FENCE
    asm volatile("" ::: "s16","s17","s18","s19","s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35","s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79","s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","s96","s97","s98","s99","s100","s101");
#endif
}

}  // namespace cnsn
