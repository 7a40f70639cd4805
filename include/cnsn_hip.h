/*
 * cnsn_hip.h — C ABI of libcnsn_hip.so: CrossNorm / SelfNorm hot path for AMD MI355X (gfx950).
 *
 * Drop-in boundary for the normalisation path of amazon-science/crossnorm-selfnorm
 * (`models/cnsn.py`).  The reference has no native layer: this header is what a binding of that
 * file's functions to native code binds to.  Each entry point names the reference interface it
 * replaces (file:line relative to the reference repository).
 *
 * Conventions
 *   - Plain C: pointers, sizes, PODs.  No C++ types, no torch types.
 *   - Every pointer is a DEVICE pointer unless it says "host".  Activations are NCHW, contiguous,
 *     16-byte aligned, element type `dtype`.  All per-plane / per-channel side arrays are float32.
 *   - The caller owns every buffer the op's entry points take (inputs, outputs, `saved`, `workspace`, `context`); those
 *     entry points enqueue their work on `stream` (a hipStream_t passed as void*; NULL = default stream), never
 *     synchronise the host and never throw.  What the library DOES own, allocate and keep — all of it outside the
 *     numerical path, listed here once (ABI 5-8; SURVEY b2 asked for none of it, DESIGN.md section 1 says why each exists):
 *       allocates   nothing on the device inside the op's calls.  The OUTPUT ARENA (cnsn_arena_*, below) allocates device
 *                   memory — hipMemCreate / hipMemMap — but only when a caller asks it for a block;
 *       synchronises  the op's calls: never.  The arena: the requesting stream when a NEW block of 384 MiB or more is timed
 *                   (~1 ms per candidate, at creation only), the streams of the blocks it releases (cnsn_arena_trim, the cap);
 *       keeps, per process   which stream the last persistent ("cluster") launch of each device went to (a launch on another
 *                   stream waits for it: one event recorded at that moment); a launch counter, two granule regions and
 *                   (ABI 8) the bases of the grid-barrier counters per exchange context — the counters themselves live in the
 *                   caller's context buffer and only grow; ONE pinned host word counting cluster launches that gave up (cnsn_resident_timeouts)
 *                   and the count forgiven by cnsn_resident_rearm; the wait bound (cnsn_set_wait_ms), the grid head-room
 *                   (cnsn_set_headroom_cus) and the strategy switches (cnsn_resident_enable); the CNSN_* environment as of
 *                   load; the arena's blocks, free lists and counters.
 *     The op's calls are safe from several host threads on different streams and devices; the setters above are
 *     process-wide and meant for start-up.
 *   - Randomness stays on the host: the batch permutation, channel permutation and boxes are
 *     INPUTS (the reference draws them with torch.randperm / numpy, models/cnsn.py:62,65,71,76).
 *   - Return value: 0 on success, <0 argument error (CNSN_E_*), >0 a hipError_t from the launch.
 *   - A box is (x1, y1, x2, y2) and selects [:, :, x1:x2, y1:y2] exactly like
 *     models/cnsn.py:66,77; x1 < 0 means "no box" (whole plane).
 */
#ifndef CNSN_HIP_H_
#define CNSN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNSN_ABI_VERSION 8

/* Largest batch whose permutation can travel as a launch argument (cnsn_problem_t.perm_host). */
#define CNSN_PERM_INLINE_MAX 1024

/* Element types of the activation tensors.  float64 is NOT offered here (the reference's eager path accepts any float tensor,
 * models/cnsn.py:12-16): CNSN_E_DTYPE.  The Python layer takes float64 tensors through these entry points on a float32 copy
 * and returns float64 (float32 accuracy in a float64 container: cnsn_amd/cnsn.py). */
enum cnsn_dtype { CNSN_F32 = 0, CNSN_BF16 = 1, CNSN_F16 = 2 };

enum cnsn_status {
    CNSN_OK = 0,
    CNSN_E_NULL = -1,       /* a required pointer is NULL                                   */
    CNSN_E_SHAPE = -2,      /* N,C,H,W not positive, or N*C*H*W does not fit in int64/limits */
    CNSN_E_DTYPE = -3,      /* unknown dtype                                               */
    CNSN_E_ALIGN = -4,      /* activation pointer not 16-byte aligned                      */
    CNSN_E_BOX = -5,        /* box outside the plane or empty                              */
    CNSN_E_WORKSPACE = -6,  /* workspace smaller than cnsn_workspace_bytes()               */
    CNSN_E_BATCH = -7,      /* SelfNorm in training mode needs N > 1 (BatchNorm1d raises)  */
    CNSN_E_STRUCT = -8,     /* cnsn_problem_t.struct_bytes does not match this library     */
    CNSN_E_UNSUPPORTED = -9 /* valid request this build cannot run (e.g. N too large)      */
};

/* Memory order of the activation tensors (ABI 6).  The reference takes any layout through `.contiguous()` (models/cnsn.py:14,16:
 * a channels-last tensor is COPIED to NCHW there).  CNSN_LAYOUT_NHWC: element (n, c, h, w) lives at ((n*H + h)*W + w)*C + c — what
 * torch.channels_last tensors and MIOpen's NHWC convolutions use; the op is computed where the tensor lies (two tensor passes each
 * way around the same per-plane algebra: for SelfNorm alone with one gate — every ResNet-50 site — ONE persistent launch per
 * direction with two grid barriers, otherwise a launch per pass; parameter gradients and running statistics as in NCHW).  `saved`
 * of a channels-last call WITHOUT CrossNorm and with one gate is the SLIM record (round 6): five floats per plane in plane
 * order + C doubles, written and read by both channels-last strategies (any forward feeds any backward of the class,
 * tests/test_gpu_saved_contract.py); cnsn_saved_floats() still answers the common record's size, an upper bound.  Un-boxed calls whose channel count
 * is a whole number of 16-byte vectors (C % 4 == 0 in fp32, C % 8 == 0 in 16 bits), no channel permutation; everything else
 * returns CNSN_E_UNSUPPORTED and the caller converts to NCHW as the reference does.  cnsn_workspace_bytes() is layout-dependent. */
enum cnsn_layout { CNSN_LAYOUT_NCHW = 0, CNSN_LAYOUT_NHWC = 1 };

enum cnsn_strategy {
    CNSN_STRATEGY_AUTO = 0,
    CNSN_STRATEGY_TWO_PASS = 1, /* stats kernel -> mid kernel -> apply kernel (3 / 5 tensor passes) */
    CNSN_STRATEGY_RESIDENT = 2, /* one launch, channel kept on chip (2 / 3 tensor passes)            */
    CNSN_STRATEGY_LOCAL = 3,    /* small planes, SelfNorm alone: a whole channel group per workgroup's LDS; falls
                                   back to two-pass where that does not apply                      */
    CNSN_STRATEGY_MONO = 4      /* small planes, SelfNorm alone: a whole channel in ONE workgroup's registers;
                                   falls back to the others where that does not apply              */
};

/* One fused CNSN.forward call (models/cnsn.py:159-164): optional CrossNorm
 * (cn_op_2ins_space_chan, :58-91) followed by optional SelfNorm (:130-150). */
typedef struct cnsn_problem {
    int32_t struct_bytes; /* = sizeof(cnsn_problem_t)                                        */
    int32_t dtype;        /* enum cnsn_dtype of x / y / grad tensors                         */
    int32_t N, C, H, W;
    /* CrossNorm part */
    int32_t cn_active;    /* 0: skip CrossNorm (module idle / eval, :104)                    */
    int32_t content_box[4]; /* crop in {content, both} (:76-78); x1<0: whole plane           */
    int32_t style_box[4];   /* crop in {style, both}   (:65-66); x1<0: whole plane           */
    float lam;            /* blend x*lam + out*(1-lam) (:86-87); 0 when lam is None          */
    float eps_cn;         /* 1e-5  (calc_ins_mean_std default, :8)                           */
    /* SelfNorm part */
    int32_t sn_active;    /* 0: skip SelfNorm (CNSN.selfnorm is None)                        */
    int32_t sn_two;       /* 1: second gate f (is_two, :123-126,:142-148)                    */
    int32_t sn_training;  /* BatchNorm1d mode: batch statistics + running-buffer update      */
    float eps_sn;         /* 1e-12 (:133)                                                    */
    float eps_bn;         /* BatchNorm1d eps, 1e-5                                           */
    float momentum;       /* BatchNorm1d momentum, 0.1                                       */
    int32_t strategy;     /* enum cnsn_strategy                                              */
    int32_t layout;       /* enum cnsn_layout: memory order of x / y / addend / grad tensors (ABI 6; was `reserved`, 0)   */
    /* Optional persistent exchange context of the cluster-resident kernels (see cnsn_context_init): device memory
     * the caller allocates ONCE per device and passes with every call.  NULL / too small: the kernels exchange
     * through `workspace`, which costs one fill launch in front of every resident launch.                       */
    void* context;
    uint64_t context_bytes;
    /* Optional (ABI 5): the batch permutation of `perm` (models/cnsn.py:62, torch.randperm on the CPU generator) as HOST
     * memory — int64 (N), read before the call returns.  The cluster-resident kernels (cnsn_which_path() ==
     * CNSN_PATH_RESIDENT, no channel permutation, N <= CNSN_PERM_INLINE_MAX) take it as a LAUNCH ARGUMENT (16-bit
     * indices): no host-to-device copy in front of the launch.  With it `perm` may be NULL; a call that resolves to
     * kernels which need the device array then returns CNSN_E_UNSUPPORTED and the caller uploads and calls again. */
    const int64_t* perm_host;
} cnsn_problem_t;

/* Parameters / buffers of one SelfNorm gate: g_fc + g_bn (or f_fc + f_bn), models/cnsn.py:118-126.
 * fc_weight is the Conv1d(C,C,k=2,groups=C) weight, shape (C,1,2) = (C,2) row-major. */
typedef struct cnsn_gate {
    const float* fc_weight; /* (C,2)                                       */
    const float* bn_weight; /* (C)                                         */
    const float* bn_bias;   /* (C)                                         */
    float* running_mean;    /* (C) updated in place when sn_training       */
    float* running_var;     /* (C) updated in place when sn_training       */
    /* (ABI 5) nn.BatchNorm1d.num_batches_tracked of the gate (models/cnsn.py:120,125: g_bn / f_bn), int64 scalar, or
     * NULL: the FORWARD adds 1 to it when sn_training — what nn.BatchNorm1d.forward does per call in training mode —
     * inside the op's own launch instead of a launch of its own.  cnsn_backward ignores it. */
    int64_t* num_batches_tracked;
} cnsn_gate_t;

typedef struct cnsn_gate_grad {
    float* d_fc_weight; /* (C,2) written (not accumulated) */
    float* d_bn_weight; /* (C)                              */
    float* d_bn_bias;   /* (C)                              */
} cnsn_gate_grad_t;

int cnsn_abi_version(void);
const char* cnsn_status_string(int status);

/* Size, in floats, of the per-plane state `cnsn_forward` writes into `saved` for `cnsn_backward`
 * (the block is opaque; internally it holds float64 scalars, so it must be 8-byte aligned). */
size_t cnsn_saved_floats(const cnsn_problem_t* prob);
/* Bytes of scratch either direction needs (the larger of the two). */
size_t cnsn_workspace_bytes(const cnsn_problem_t* prob);

/* Fused forward.  Replaces CNSN.forward -> CrossNorm.forward -> cn_op_2ins_space_chan ->
 * instance_norm_mix -> calc_ins_mean_std and SelfNorm.forward (models/cnsn.py:159-164, 103-110,
 * 58-91, 20-29, 8-17, 130-150) for one activation tensor.
 *   perm       int64 (N)  batch permutation (:62); required when cn_active
 *   chan_perm  int64 (C)  channel permutation (:71) or NULL (chan=False)
 *   g, f       gates; g required when sn_active, f when sn_two
 *   y          output, same shape/dtype as x, must not alias x
 *   saved      cnsn_saved_floats() floats, or NULL when no backward will follow */
int cnsn_forward(const cnsn_problem_t* prob, const void* x, const int64_t* perm,
                 const int64_t* chan_perm, const cnsn_gate_t* g, const cnsn_gate_t* f, void* y,
                 float* saved, void* workspace, size_t workspace_bytes, void* stream);

/* Fused backward: what autograd derives for the forward above (gradient flows through the content
 * statistics, the permuted style statistics — the style source is not detached, :66-68 — and the
 * SelfNorm gate incl. BatchNorm1d batch statistics).  grad_x must not alias grad_y or x. */
int cnsn_backward(const cnsn_problem_t* prob, const void* grad_y, const void* x,
                  const int64_t* perm, const int64_t* chan_perm, const cnsn_gate_t* g,
                  const cnsn_gate_t* f, const float* saved, void* grad_x,
                  const cnsn_gate_grad_t* dg, const cnsn_gate_grad_t* df, void* workspace,
                  size_t workspace_bytes, void* stream);

/* ---- residual-block epilogue fused around the op -------------------------------
 * The callers wrap CNSN in element-wise neighbours that each cost a full tensor pass:
 *   models/imagenet/resnet_cnsn.py:117-122  out += identity ; out = cnsn(out) ; out = relu(out)    (pos='post')
 *   models/imagenet/resnet_cnsn.py:112-122  out = cnsn(out) ; out += identity ; relu               (pos='residual',
 *                                            and 'identity' with the roles of the two tensors swapped)
 *   models/cifar/wideresnet_cnsn.py:93-96   out = torch.add(x, out) ; return cnsn(out)             (pos='post')
 * cnsn_forward_fused / cnsn_backward_fused evaluate  y = act( CNSN(x [+ addend]) [+ addend] )  in the
 * same launches as the op itself: the sum is formed in registers on the way in (PRE) or on the way
 * out (POST), the ReLU on the way out; the backward recomputes the ReLU mask from the forward
 * coefficients kept in `saved` instead of reading y. */
enum cnsn_add_mode {
    CNSN_ADD_NONE = 0,
    CNSN_ADD_PRE = 1, /* CNSN input is x + addend                     */
    CNSN_ADD_POST = 2 /* addend is added to CNSN's output            */
};

typedef struct cnsn_epilogue {
    int32_t struct_bytes; /* = sizeof(cnsn_epilogue_t)                                        */
    int32_t add_mode;     /* enum cnsn_add_mode                                               */
    int32_t relu;         /* 1: y = max(y, 0) last (nn.ReLU, resnet_cnsn.py:122)              */
    int32_t reserved;
    const void* addend;   /* same shape / dtype / layout as x; NULL iff add_mode == NONE       */
    void* sum_out;        /* (ABI 8) add_mode PRE only, may be NULL: where X = x + addend is KEPT (x's shape / dtype /
                           * layout; rounded to the dtype like the reference's in-place `out += identity`,
                           * resnet_cnsn.py:117; it must NOT alias x or addend: every pixel chunk of a plane reads the
                           * plane's first pixel as its shift).  A forward that
                           * wrote it is followed by cnsn_backward_fused with add_mode NONE (same relu), x = sum_out and
                           * the forward's `saved`: the backward then reads ONE tensor instead of two, twice — the
                           * channels-last strategies (two tensor passes each way) move 10 tensor passes per step
                           * instead of 12, and the forward's second pass finds X in the caches it was just written
                           * through.  Only calls for which cnsn_keeps_sum() answers 1 take it; others return
                           * CNSN_E_UNSUPPORTED when it is set (single-touch strategies gain nothing from it). */
} cnsn_epilogue_t;

/* 1: cnsn_forward_fused(prob, epi, ...) writes epi->sum_out when it is set (channels-last layout, add_mode PRE, a call the
 * channels-last kernels take); 0 otherwise.  (ABI 8) */
int cnsn_keeps_sum(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi);

/* As cnsn_forward with the epilogue `epi` (NULL = none).  `saved` from this call must go to
 * cnsn_backward_fused with the same `epi`. */
int cnsn_forward_fused(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const void* x,
                       const int64_t* perm, const int64_t* chan_perm, const cnsn_gate_t* g,
                       const cnsn_gate_t* f, void* y, float* saved, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Backward of cnsn_forward_fused.  grad_x is the gradient of x.
 *   add_mode PRE : the gradient of addend equals grad_x (same values) — the caller aliases it.
 *   add_mode POST: the gradient of addend is grad_y masked by the ReLU; it is written to
 *                  grad_addend when relu == 1 (required then), and equals grad_y otherwise
 *                  (grad_addend ignored, may be NULL). */
int cnsn_backward_fused(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const void* grad_y,
                        const void* x, const int64_t* perm, const int64_t* chan_perm,
                        const cnsn_gate_t* g, const cnsn_gate_t* f, const float* saved, void* grad_x,
                        void* grad_addend, const cnsn_gate_grad_t* dg, const cnsn_gate_grad_t* df,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- the NEXT block's BatchNorm2d + ReLU behind the op (second half) -------------------------
 * WideResNet ends a block with `out = torch.add(x, out); return self.cnsn(out)` (models/cifar/wideresnet_cnsn.py:93-96)
 * and starts the next one with `relu1(bn1(.))` on that tensor (:69-70 when the widths differ, :76-77 when they are
 * equal — then the un-normalised tensor is ALSO the next block's shortcut, :93; after the last block: :222).  These entry
 * points evaluate   y = CNSN(x [+ addend]),  z = relu(BatchNorm2d(y))   in the op's own launch: the channel's plane
 * moments are on chip, y is a per-plane multiple of its input, so BatchNorm2d's batch statistics over (N, H, W) follow
 * algebraically and z is one more store stream; the backward takes the gradients of BOTH outputs.
 *   - SelfNorm alone (cn_active = 0, one gate); epilogue: add_mode NONE or PRE, relu = 0.  Anything else, and shapes
 *     no fused kernel covers (cnsn_bnrelu_plan() == 0), return CNSN_E_UNSUPPORTED: the caller then issues
 *     cnsn_forward_fused and its own BatchNorm2d + ReLU, which is what the reference does.
 *   - y / grad_y may be NULL: the next block consumes only z when its widths differ (:69-70) and after the last block.
 *   - bn_stats: float32 (2, C) — the batch mean and rstd BatchNorm2d normalised with; forward writes, backward reads.
 *   - `saved` from the forward goes to the backward together with the same `tail` parameters (weight unchanged). */
typedef struct cnsn_bn_tail {
    int32_t struct_bytes;   /* = sizeof(cnsn_bn_tail_t)                                              */
    int32_t training;       /* nn.BatchNorm2d mode: 1 = batch statistics + running-buffer update       */
    float eps;              /* 1e-5                                                                  */
    float momentum;         /* 0.1                                                                   */
    const float* weight;    /* (C)                                                                   */
    const float* bias;      /* (C)                                                                   */
    float* running_mean;    /* (C) updated in place when training                                    */
    float* running_var;     /* (C) (takes the unbiased batch variance, like torch)                   */
    int64_t* num_batches_tracked; /* (ABI 5) int64 scalar or NULL: += 1 by the forward when training          */
} cnsn_bn_tail_t;

/* 1 when a fused kernel takes the call (pure function of the problem), 0 when not, < 0 argument error */
int cnsn_bnrelu_plan(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, int backward);
int cnsn_forward_bnrelu(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const cnsn_bn_tail_t* tail,
                        const void* x, const cnsn_gate_t* g, void* y, void* z, float* saved, float* bn_stats,
                        void* workspace, size_t workspace_bytes, void* stream);
/* d_bn_weight / d_bn_bias: float32 (C), written (not accumulated) */
int cnsn_backward_bnrelu(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const cnsn_bn_tail_t* tail,
                         const void* grad_y, const void* grad_z, const void* x, const cnsn_gate_t* g,
                         const float* saved, const float* bn_stats, void* grad_x, const cnsn_gate_grad_t* dg,
                         float* d_bn_weight, float* d_bn_bias, void* workspace, size_t workspace_bytes, void* stream);

/* calc_ins_mean_std (models/cnsn.py:8-17): mean and sqrt(unbiased var + eps) of every (n,c)
 * plane, optionally of a box of it (the reference takes the box by slicing, :66,:77).
 * mean_std: float32 (2, N*C) — row 0 the means, row 1 the stds. */
int cnsn_plane_stats(const void* x, int dtype, int N, int C, int H, int W, const int32_t* box,
                     float eps, float* mean_std, void* stream);

/* Backward of cnsn_plane_stats: dx = dmean/M + dstd*(x-mean)/(std*(M-1)) inside the box, 0 outside. */
int cnsn_plane_stats_backward(const void* x, int dtype, int N, int C, int H, int W,
                              const int32_t* box, const float* mean, const float* std,
                              const float* dmean, const float* dstd, void* dx, void* stream);

/* y = scale[p] * x + shift[p] per plane p=(n,c): the re-normalise of instance_norm_mix
 * (models/cnsn.py:27-29) and SelfNorm's x*g (:150) once the coefficients are known. */
int cnsn_plane_affine(const void* x, int dtype, int N, int C, int H, int W, const float* scale,
                      const float* shift, void* y, void* stream);

/* Per-plane sums needed by the backward of cnsn_plane_affine.
 * sums: float32 (2, N*C) — row 0 sum(g), row 1 sum(g*x) of every plane. */
int cnsn_plane_dot(const void* g, const void* x, int dtype, int N, int C, int H, int W,
                   float* sums, void* stream);

/* The two launches of a dedicated InstanceNorm2d backward (models/imagenet/resnet_ibn_cnsn.py:24-44 — what autograd
 * derives for nn.InstanceNorm2d): 5 tensor passes instead of the 9 of composing the two functions above.
 *   cnsn_plane_dot_shifted : sums (2, N*C) — row 0 sum(g), row 1 sum(g * (x - float(shift[p]))) — `shift` float64 (N*C)
 *                            (the plane means: the second sum is formed about the mean, no cancellation afterwards)
 *   cnsn_plane_combine     : out = cG[p]*g + cX[p]*(x - xr[p]) + c0[p];  coef float32 (4, N*C): rows cG, cX, xr, c0 */
int cnsn_plane_dot_shifted(const void* g, const void* x, int dtype, int N, int C, int H, int W,
                           const double* shift, float* sums, void* stream);
int cnsn_plane_combine(const void* g, const void* x, int dtype, int N, int C, int H, int W, const float* coef,
                       void* out, void* stream);

/* Which kernels a call would run (pure function of the problem; nothing is launched): lets a caller, a test or a
 * benchmark see what CNSN_STRATEGY_AUTO — or a forced strategy with its fall-backs — resolves to. */
enum cnsn_path {
    CNSN_PATH_STREAMING = 0, /* two-pass, 16/64/256 lanes per plane                       */
    CNSN_PATH_PACKED = 1,    /* two-pass, runs of small planes staged through LDS         */
    CNSN_PATH_RESIDENT = 2,  /* one launch, cluster of workgroups per channel             */
    CNSN_PATH_LOCAL = 3,     /* one launch, a whole channel group per workgroup (LDS)     */
    CNSN_PATH_MONO = 4       /* one launch, a whole channel per workgroup (registers)     */
};
int cnsn_which_path(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, int has_chan_perm, int backward);
/* Refinement of CNSN_PATH_RESIDENT: 1 when the call runs the partial-moment cluster kernels (SelfNorm alone; since ABI 5 also the
 * backward of un-boxed CrossNorm + SelfNorm) (models/cnsn.py:130-150 with
 * no CrossNorm armed — every ResNet-50 / WideResNet site on an idle step): the workgroups of a channel exchange PARTIAL
 * batch moments of BatchNorm1d's input (:121,138) instead of every plane's statistics.  0 otherwise, < 0 argument error. */
int cnsn_sn_cluster_plan(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, int backward);

/* ---- the block's LAST BatchNorm2d in front of the op (round 6, ABI 8; SURVEY §8 f1 widened by one neighbour) ---------------
 * A ResNet bottleneck ends `out = self.bn3(out); out += identity; out = self.cnsn(out); out = self.relu(out)`
 * (models/imagenet/resnet_cnsn.py:108-122, pos='post').  These entry points evaluate
 *       y = act( CNSN( BatchNorm2d(conv_out) + identity ) )
 * in ONE persistent launch per direction: BatchNorm2d's batch statistics and the statistics of the sum follow from one set of
 * per-plane sums of conv_out and identity, neither bn3's output nor the sum is ever written, and the backward re-evaluates
 * them from the two tensors BatchNorm2d and the add would have saved anyway: 13 tensor passes per block and step instead of
 * 8 (BatchNorm2d) + 10 (the op).  Values are rounded where the un-fused sequence rounds them (bn3's output, the sum, y); the
 * statistics are those of the un-rounded sum (rounding noise of zero mean: within north_star's tolerances).
 *   - channels-last layout, SelfNorm alone (cn_active = 0, one gate), training mode in BOTH normalisations, N <= 256;
 *     epilogue: add_mode PRE with addend = identity (sum_out NULL), relu 0 / 1.  Anything else — cnsn_bn_block_plan() == 0 —
 *     returns CNSN_E_UNSUPPORTED: the caller then runs BatchNorm2d itself and cnsn_forward_fused, as the reference does.
 *   - bn: BatchNorm2d's parameters and buffers (the struct of the fused tail); running statistics and num_batches_tracked are
 *     updated by the forward as nn.BatchNorm2d does (momentum; running_var takes the unbiased variance).
 *   - bn_skip (may be NULL): the skip path ends in a BatchNorm2d of its own — the block's `downsample` (resnet_cnsn.py:99-100,
 *     the first block of every stage).  epi->addend is then the INPUT of that BatchNorm2d (the 1x1 convolution's output) and
 *       y = act( CNSN( BatchNorm2d(conv_out) + BatchNorm2d_skip(addend) ) ):
 *     the sum is affine in both convolution outputs per channel, the same plane sums serve both normalisations, and the
 *     backward writes the gradient of the skip convolution's output into grad_identity.  Same tensor passes: the downsample's
 *     BatchNorm2d costs nothing.
 *   - bn_stats: float32 (4, C) — batch mean, rstd and the two coefficients x = alpha*conv_out + beta was evaluated with
 *     ((8, C) with bn_skip: its four rows follow); forward writes, backward reads.  `saved`: cnsn_saved_floats(), as for
 *     cnsn_forward_fused; workspace: cnsn_workspace_bytes().
 *   - backward: grad_conv_out and grad_identity (the gradient of the sum; with bn_skip: of the skip convolution's output) are
 *     both written; d_bn_weight / d_bn_bias (and d_bn_skip_*): (C). */
int cnsn_bn_block_plan(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi);
int cnsn_forward_bn_block(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const cnsn_bn_tail_t* bn,
                          const cnsn_bn_tail_t* bn_skip, const void* conv_out, const cnsn_gate_t* g, void* y, float* saved,
                          float* bn_stats, void* workspace, size_t workspace_bytes, void* stream);
int cnsn_backward_bn_block(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const cnsn_bn_tail_t* bn,
                           const cnsn_bn_tail_t* bn_skip, const void* grad_y, const void* conv_out, const cnsn_gate_t* g,
                           const float* saved, const float* bn_stats, void* grad_conv_out, void* grad_identity,
                           const cnsn_gate_grad_t* dg, float* d_bn_weight, float* d_bn_bias, float* d_bn_skip_weight,
                           float* d_bn_skip_bias, void* workspace, size_t workspace_bytes, void* stream);

/* ---- persistent exchange context of the cluster-resident strategy --------------------------------
 * The resident kernels hand per-plane scalars from workgroup to workgroup through device memory.  Through the
 * per-call `workspace` (contents unknown) that memory has to be filled with an 'empty' pattern by a launch of its own
 * before every resident launch.  A CONTEXT is memory whose contents only this library ever writes: every value in
 * it carries the number of the launch that wrote it, so nothing is ever cleared (the library counts launches per
 * context on the host; on wrap-around it clears the context once, stream-ordered).
 *   cnsn_context_bytes(prob)  bytes a context must have for `prob` to use it (0: this problem never would)
 *   cnsn_context_init(...)    zero-fills a context on `stream` and forgets its launch count; REQUIRED once for
 *                             every buffer before it is passed as cnsn_problem_t.context; the caller orders the
 *                             fill before the first use (same stream, or a synchronisation)
 * Its last 4 MiB are two regions of untagged granules that launches on large tensors use alternately, each launch
 * clearing the other region for the next one (no fill launch; cnsn_context_bytes includes them).  In front of them (ABI 8):
 * a 4 KiB BARRIER BLOCK — the arrival counters of the single-launch channels-last kernels' grid barrier (eight group counters,
 * a top counter, eight generation words, each in a cache line of its own).  They only grow; the library keeps, per context,
 * what the launches so far have left in them, so nothing is cleared between launches (a launch that gave up leaves them
 * short: the next launch that needs the block zeroes it, stream-ordered).
 * One context per device, used by one stream at a time or by streams the library orders itself (its per-device
 * launch chaining).  Not used while `stream` is being captured into a graph (a replay would repeat the launch
 * number): such calls fall back to the workspace.  Nothing in the reference corresponds to this. */
size_t cnsn_context_bytes(const cnsn_problem_t* prob);
int cnsn_context_init(void* context, size_t bytes, void* stream);

/* ---- health of the cluster-resident strategy ----------------------------------------------------
 * The resident kernels exchange per-plane scalars between co-resident workgroups with bounded spin waits.  When a
 * wait runs out (seconds: the GPU is shared with something that keeps part of the grid off the device) the launch
 * gives up WITHOUT trapping: its outputs are incomplete, this counter goes up, and CNSN_STRATEGY_AUTO stops choosing
 * the resident kernels for the rest of the process (two-pass from then on).  A caller polls the counter — a plain
 * host read, no synchronisation — and repeats the step that was in flight.  Nothing in the reference corresponds to
 * this (eager PyTorch has no cross-workgroup exchange); it belongs to the replacement of models/cnsn.py:159-164. */
int cnsn_resident_timeouts(void);
/* Re-arm (ABI 6): forgive the time-outs counted so far, so that the resident kernels may be chosen again (by
 * CNSN_STRATEGY_AUTO once cnsn_resident_enable(1) has been called too).  The cause of a time-out is usually transient —
 * another process's kernel held part of the GPU for seconds — and a job that runs for days should not pay the two-pass
 * kernels for the rest of its life: the training loops re-arm after a number of clean steps that doubles with every
 * relapse (callers/steps.py::StepGuard, the same count on every data-parallel rank).  cnsn_resident_timeouts() keeps
 * counting from where it was.  Returns how often this process has re-armed.  cnsn_resident_degraded(): 1 while a
 * time-out is unforgiven. */
int cnsn_resident_rearm(void);
int cnsn_resident_degraded(void);
/* on = 0: CNSN_STRATEGY_AUTO never chooses the resident kernels (same as the environment variable CNSN_RESIDENT=0
 * at load time); on = 1: allowed again.  CNSN_STRATEGY_RESIDENT is not affected. */
void cnsn_resident_enable(int on);

/* Bound of a cluster wait in milliseconds (default 5000; the environment variable CNSN_WAIT_MS at load time).  A process
 * that is one rank of a data-parallel job lowers it (its peers wait with it in the next collective): the Python layer
 * sets 2000 under an initialised process group of more than one rank.  ms <= 0: back to the default.  A CNSN_WAIT_MS
 * knob in force (at load time or after cnsn_reload_env) takes precedence over the value set here. */
void cnsn_set_wait_ms(int ms);
int cnsn_wait_ms(void); /* the bound in force, in milliseconds */

/* Grid head-room (ABI 7): the persistent ("cluster") grids are sized for `n` compute units fewer than the part has (n <= 0:
 * the whole part).  Next to a kernel that HOLDS compute units for longer than a launch — RCCL's channel kernels during a
 * large all-reduce — a full-size persistent grid still completes, in order, but takes 1.5-1.7 x its quiet time; a grid sized for
 * what is free takes its quiet time (profiles/r05_exchange_hardening.md).  The Python layer sets it under an initialised
 * process group of more than one rank (two compute units per RCCL channel, NCCL_MAX_NCHANNELS or 16 of them).  A
 * CNSN_HEADROOM_CUS knob in force takes precedence.  Nothing in the reference corresponds to this. */
void cnsn_set_headroom_cus(int n);
int cnsn_headroom_cus(void); /* the head-room in force */

/* ---- output arena (ABI 6; torch-visible since ABI 7) ------------------------------------------------
 * The reference's op returns NEW tensors (models/cnsn.py:29 `return (...) * style_std + style_mean`, :150 `return x * g`),
 * which torch's caching allocator places.  On MI355X the single-touch launches' plane-strided writes run 10-20 % slower
 * into four out of five large hipMalloc'ed blocks than into the fifth (profiles/r04_memory_map.md), and no allocator
 * option chooses.  The arena creates blocks it can choose among: address ranges mapped from physical allocations of its
 * own (hipMemCreate chunks, default 56 MiB, environment CNSN_ARENA_CHUNK_MB, plus one tail chunk — the chunk size does NOT
 * decide a block's speed, profiles/r05_arena.md); a NEW block of 384 MiB or more is the fastest of cnsn_arena_set_tries()
 * candidates created together and timed with a plane-strided fill on the requesting stream (which is synchronised, ~1 ms per
 * candidate; not while the stream is being captured), the others' memory handed back at once.
 * The library itself still takes caller-owned output pointers everywhere — a caller MAY get them here.
 *
 * Two ways in.
 *  (1) cnsn_arena_map / cnsn_arena_unmap (ABI 7): create / release ONE block; nothing is cached by the library.  The pair has
 *      the signature of a torch pluggable allocator (alloc(size, device, stream), free(ptr, size, device, stream)): the Python
 *      layer's arena is a `torch.cuda.MemPool` over it, so the caching allocator caches, splits, counts
 *      (`memory_allocated`), orders across streams (`record_stream`) and — out of memory anywhere in the process — releases
 *      these blocks like its own.  cnsn_arena_map returns NULL when the device has no memory (torch then frees its caches and
 *      asks again) or the driver has no virtual-memory support (remembered per device).  cnsn_arena_unmap gives the physical
 *      memory back; the caller has ordered it behind every use of the block.
 *  (2) cnsn_arena_alloc / cnsn_arena_free: the same blocks behind a caching layer of the library's own, for callers without an
 *      allocator to plug into.
 *   cnsn_arena_alloc   device memory of at least `bytes` (rounded up to the driver's granularity, 2 MiB) on `device`, to be
 *                      used first on `stream`.  Served from the free lists when a block of that size — or the smallest one
 *                      within an eighth above it — is free (a mutex and a list operation; a block last used on other streams
 *                      is ordered behind their queued work); else a new block is created, after evicting the least recently
 *                      used free blocks if the arena would exceed its cap (cnsn_arena_set_limit).  When the device is out of
 *                      memory every free block is released and the request repeated once.  NULL: no memory, cap reached with
 *                      everything in use, or no virtual-memory support — the caller allocates as it always did.
 *   cnsn_arena_free    give a block back (pointer as returned by cnsn_arena_alloc); work already queued on the block's
 *                      stream may still use it — the next user is ordered behind it, like a caching allocator's.
 *   cnsn_arena_record_stream  `ptr` (from cnsn_arena_alloc, not yet freed) is also used on `stream`: whoever gets the block
 *                      next is ordered behind that stream's work as well (Tensor.record_stream's contract).
 *   cnsn_arena_set_limit  cap in bytes on what the caching layer holds on `device` (in use + free); 0: back to the default
 *                      (environment CNSN_ARENA_MAX_MB, else half of the device memory).  Free blocks above a new cap are
 *                      released at once.
 *   cnsn_arena_trim    give the physical memory of every FREE block of `device` (-1: all devices) back to the driver;
 *                      returns the bytes released.  Synchronises the streams those blocks were last used on.  The
 *                      blocks' ADDRESS ranges stay reserved for the life of the process: a range that was unmapped is
 *                      never mapped again (on ROCm 7.2 a re-used range was accessed through stale translations).
 *   cnsn_arena_prospect  look for fast memory, explicitly and bounded: create `candidates` blocks for `bytes` (all alive
 *                      at once; never more than half of the free device memory), time a plane-strided fill into each
 *                      (~1 ms per block on `stream`, which is synchronised), leave the `keep` fastest on the free list
 *                      and give the others back.  Returns the number kept.  `gbps_out` (candidates floats, optional)
 *                      receives every candidate's measured write rate.  WHERE a block lies physically decides its
 *                      write rate (about one block in five takes plane-strided writes 15-20 % faster,
 *                      profiles/r04_memory_map.md); nothing calls this by default.
 *   cnsn_arena_block_gbps  the rate measured for the block that contains `ptr` (0: never measured).
 *   cnsn_arena_owns    1 when `ptr` lies inside a block of the arena (either way in).
 * Not to be used while the stream is being captured into a graph (a replay needs addresses nobody else re-uses). */
typedef struct cnsn_arena_stats {
    int32_t struct_bytes; /* = sizeof(cnsn_arena_stats_t) */
    int32_t device;
    uint64_t chunk_bytes;   /* size of one physical allocation                       */
    uint64_t mapped_bytes;  /* physical memory held by the arena on this device (both ways in) */
    uint64_t in_use_bytes;  /* ... of which handed out (a mapped block counts as handed out to its cache) */
    uint64_t blocks, blocks_in_use;
    uint64_t hits;          /* requests served from the free list                    */
    uint64_t misses;        /* requests that created and mapped a new block          */
    uint64_t failed;        /* requests answered with NULL                           */
    uint64_t probed;        /* candidate blocks timed when blocks were created       */
    uint64_t tries;         /* candidates per new block in force (0: not resolved)   */
    uint64_t evicted;       /* free blocks released because of the cap / a full device (ABI 7) */
    uint64_t limit_bytes;   /* the caching layer's cap in force (0: not resolved yet)  (ABI 7) */
    uint64_t broken;        /* 1: the driver refused the virtual-memory calls on this device (ABI 7) */
} cnsn_arena_stats_t;
void* cnsn_arena_map(size_t bytes, int device, void* stream);
void cnsn_arena_unmap(void* ptr, size_t bytes, int device, void* stream);
void* cnsn_arena_alloc(int device, size_t bytes, void* stream);
int cnsn_arena_free(void* ptr);
int cnsn_arena_record_stream(void* ptr, void* stream);
int cnsn_arena_set_limit(int device, uint64_t bytes);
size_t cnsn_arena_trim(int device);
int cnsn_arena_owns(const void* ptr);
int cnsn_arena_stats(int device, cnsn_arena_stats_t* out);
int cnsn_arena_prospect(int device, size_t bytes, int keep, int candidates, void* stream, float* gbps_out);
/* candidates per NEW block of 384 MiB or more (default 8, environment CNSN_ARENA_TRIES; 1: take what the driver gives): a block
 * is chosen as the fastest of that many created together and timed with the plane-strided fill, the others handed back at once
 * — the arena's standing policy, paid when a block is created (the first steps of a job).  Smaller blocks are never timed: the Infinity Cache absorbs a write of that size.
 * Returns the previous value; < 1: back to the environment's / default. */
int cnsn_arena_set_tries(int tries);
int cnsn_arena_block_gbps(const void* ptr, float* gbps);
/* chunk size for blocks created from now on (0: back to CNSN_ARENA_CHUNK_MB / 56 MiB); blocks of another chunk size stay
 * valid and on their free lists but no longer serve requests (cnsn_arena_trim releases them).  A measurement knob
 * (tools/arena_probe.py, profiles/r05_arena.md). */
int cnsn_arena_set_chunk_bytes(size_t chunk_bytes);

/* ---- environment knobs ---------------------------------------------------------------------------
 * The library's CNSN_* environment variables (tuning and test switches: CNSN_WAIT_MS, CNSN_RESIDENT, CNSN_PIPE, ... —
 * the list is csrc/cnsn_env.h) are read ONCE, when the library is loaded; no launch calls getenv().  This re-reads
 * them, for tests and A/B tools that change a knob inside one process.  Not to be called while another thread is
 * inside the library.  Nothing in the reference corresponds to this. */
void cnsn_reload_env(void);

/* ---- Jensen-Shannon consistency of three views ----------------------------------
 * imagenet.py:367-381 / cifar.py:173-186: p_i = softmax(logits_i, 1); lm = clamp(mean_i p_i, 1e-7, 1).log();
 * loss = mean_i F.kl_div(lm, p_i, reduction='batchmean').  One launch computes the loss and, when the three
 * gradient pointers are given, d loss / d logits_i (same dtype as the logits; scale by the upstream gradient
 * of the scalar loss on the caller's side).  logits: (B, K) row-major, contiguous.
 *   loss       float32 scalar (device)
 *   workspace  cnsn_jsd_workspace_bytes(B) bytes */
size_t cnsn_jsd_workspace_bytes(int B);
int cnsn_jsd(const void* logits_clean, const void* logits_aug1, const void* logits_aug2, int dtype, int B,
             int K, float* loss, void* d_clean, void* d_aug1, void* d_aug2, void* workspace,
             size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CNSN_HIP_H_ */
