// Environment knobs of the library (CNSN_*), read ONCE when the library is loaded.
//
// Every knob is a tuning / test switch; none is needed for normal use.  Until round 3 each launch called getenv() for the
// knobs on its path (8 or more per call): the values are now copied into a table by a static constructor at dlopen and
// the launch paths read the table.  `cnsn_reload_env()` (include/cnsn_hip.h) re-reads the environment — for tests and
// A/B tools that flip a knob inside one process; it must not run concurrently with launches.
#pragma once

namespace cnsn {

enum Knob {
    K_LOCAL_LB = 0,   // CNSN_LOCAL_LB      threads per workgroup of the channel-local kernels (256 | 1024)
    K_LOCAL_CG,       // CNSN_LOCAL_CG      channels per workgroup of the channel-local kernels
    K_STAGGER,        // CNSN_STAGGER       start-up stagger of the cluster kernels (tuning)
    K_WAIT_MS,        // CNSN_WAIT_MS       bound of a cluster wait in milliseconds
    K_FAULT_INJECT,   // CNSN_FAULT_INJECT  1: one cluster member never publishes (tests of the give-up path)
    K_DEBUG,          // CNSN_DEBUG         print the resolved grids
    K_PROF,           // CNSN_PROF          phase stamps (CNSN_PROF builds only)
    K_MONO,           // CNSN_MONO          0: AUTO never takes the channel-in-registers kernels
    K_MONO_RELOAD,    // CNSN_MONO_RELOAD   0: no part-wise backward (A/B runs)
    K_NO_PACKED,      // CNSN_NO_PACKED     1: no packed two-pass kernels
    K_MID_TILE,       // CNSN_MID_TILE      1: one channel per workgroup in the mid kernels, 8: tiles of 8 channels only (A/B)
    K_SNX,            // CNSN_SNX           SelfNorm-only cluster kernels: 0 never, 1 AUTO rule, 2 wherever instantiated
    K_RESIDENT,       // CNSN_RESIDENT      0: AUTO never takes a cluster-resident kernel
    K_CONTEXT,        // CNSN_CONTEXT       0: exchange through the workspace, 1: through the context at every size
    K_EPOCH_START,    // CNSN_EPOCH_START   first launch number of a context (tests of the wrap-around)
    K_KEEP,           // CNSN_KEEP          cache policy of the two-pass kernels' first pass
    K_PIPE,           // CNSN_PIPE          pipelined cluster kernels: 0 never, 1 AUTO rule, 2 wherever instantiated
    K_WIDE,           // CNSN_WIDE          channel-group kernels: 0 never, 1 AUTO rule, 2 wherever eligible
    K_PONG,           // CNSN_PONG          granule regions: 0 never (fill launches), 2 also for small tensors (tests)
    K_SNXCN,          // CNSN_SNXCN         CrossNorm-capable partial-moment backward: 0 never, 1 AUTO rule, 2 wherever instantiated
    K_ARENA_CHUNK_MB, // CNSN_ARENA_CHUNK_MB  size of the output arena's physical allocations in MiB (default 56)
    K_XCD,            // CNSN_XCD           1: a cluster's workgroups share ONE XCD (A/B knob; default: consecutive workgroups, all 8 XCDs)
    K_HEADROOM_CUS,   // CNSN_HEADROOM_CUS  compute units the persistent grids leave to others (RCCL's channel kernels): default 0
    K_ARENA_TRIES,    // CNSN_ARENA_TRIES   candidates the output arena times per new block of 384 MiB or more (default 8; 1: none)
    K_ARENA_SPREAD_GB,// CNSN_ARENA_SPREAD_GB  GB of physical memory held between an arena block's candidates (A/B knob, default 0)
    K_MID_BLOCK,      // CNSN_MID_BLOCK     256: the mid kernels never take 1024-thread workgroups (A/B knob)
    K_NHWC_FUSED,     // CNSN_NHWC_FUSED    single-launch channels-last kernels: 0 never, 1 AUTO (default), 2 wherever they apply, n > 2: AUTO up to n MiB
    K_ARENA_MAX_MB,   // CNSN_ARENA_MAX_MB  cap in MiB on what cnsn_arena_alloc's cache holds per device (default: half of the device memory)
    K_COUNT
};

// value of the knob as of load (or the last cnsn_reload_env), nullptr when unset; the pointer stays valid until a reload
const char* knob(Knob k);
void reload_knobs();

}  // namespace cnsn
