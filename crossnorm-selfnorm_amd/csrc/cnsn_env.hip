// The CNSN_* environment knobs, snapshotted at load (cnsn_env.h).
#include "cnsn_env.h"

#include <cstdlib>
#include <mutex>
#include <string>

namespace cnsn {
namespace {

const char* const kNames[K_COUNT] = {
    "CNSN_LOCAL_LB", "CNSN_LOCAL_CG", "CNSN_STAGGER",     "CNSN_WAIT_MS", "CNSN_FAULT_INJECT", "CNSN_DEBUG",
    "CNSN_PROF",     "CNSN_MONO",     "CNSN_MONO_RELOAD", "CNSN_NO_PACKED", "CNSN_MID_TILE",   "CNSN_SNX",
    "CNSN_RESIDENT", "CNSN_CONTEXT",  "CNSN_EPOCH_START", "CNSN_KEEP",    "CNSN_PIPE",         "CNSN_WIDE",
    "CNSN_PONG",
    "CNSN_SNXCN",
    "CNSN_ARENA_CHUNK_MB",
    "CNSN_XCD",
    "CNSN_HEADROOM_CUS",
    "CNSN_ARENA_TRIES",
    "CNSN_ARENA_SPREAD_GB",
    "CNSN_MID_BLOCK",
    "CNSN_NHWC_FUSED",
    "CNSN_ARENA_MAX_MB",
};

struct Table {
    std::string text[K_COUNT];
    const char* value[K_COUNT];
    std::mutex mu;
    Table() { read(); }
    void read() {
        std::lock_guard<std::mutex> lock(mu);
        for (int k = 0; k < K_COUNT; ++k) {
            const char* e = getenv(kNames[k]);
            if (e) {
                text[k] = e;
                value[k] = text[k].c_str();
            } else {
                text[k].clear();
                value[k] = nullptr;
            }
        }
    }
};

Table& table() {
    static Table t;  // (also reachable from other translation units' static constructors)
    return t;
}

struct AtLoad {
    AtLoad() { (void)table(); }
} g_at_load;  // dlopen runs this: the environment is read when the library is loaded, not at the first launch

}  // namespace

const char* knob(Knob k) { return table().value[k]; }
void reload_knobs() { table().read(); }

}  // namespace cnsn
