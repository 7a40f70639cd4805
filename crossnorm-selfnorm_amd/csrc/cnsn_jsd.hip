// Jensen-Shannon consistency loss of three views, forward and gradient in one launch.
//
// Reference arithmetic (imagenet.py:367-381, cifar.py:173-186):
//     p_i   = softmax(logits_i, dim=1)                       i = clean, aug1, aug2
//     lm    = clamp((p_0 + p_1 + p_2) / 3, 1e-7, 1).log()
//     loss  = ( KL(lm || p_0) + KL(lm || p_1) + KL(lm || p_2) ) / 3,   KL = F.kl_div(lm, p_i, 'batchmean')
//           = 1/(3B) * sum_rows sum_i sum_k p_ik (log p_ik - lm_k)
// which eager torch spreads over ~20 small launches forward and as many backward on three (B, K) tensors (K = 100
// or 1000): pure launch latency.  Here one workgroup owns one row of the three tensors: three passes over
// 3K values that stay in L1/L2 (row maxima and exp-sums; loss terms and the softmax-backward dot products; the
// gradient), then a second, single-workgroup launch adds the B row losses in a fixed order (deterministic).
//
// Gradient (what autograd derives, F.kl_div being differentiable in input AND target):
//     g_ik  = dL/dp_ik = c (log p_ik - lm_k)            where 1e-7 <= m_k <= 1 (the -1 from d lm/dp cancels the +1)
//                      = c (log p_ik + 1 - lm_k)        where the clamp is active (lm_k constant)
//     dL/dz_ik = p_ik (g_ik - sum_k' p_ik' g_ik'),   c = 1/(3B)
// A probability that underflowed to exactly 0 contributes 0 to the loss (torch: xlogy) and gets gradient 0 here
// (eager torch produces NaN there: 0/0 in xlogy's backward).
#include "../../include/cnsn_hip.h"

#include <hip/hip_runtime.h>

#include "cnsn_device.h"
#include "cnsn_host_common.h"

using namespace cnsn;

namespace {

__device__ __forceinline__ float block_max(float v, float* lds) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(lds[0], lds[1]), fmaxf(lds[2], lds[3]));
}
__device__ __forceinline__ float block_sum(float v, float* lds) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    return (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void jsd_rows_kernel(const T* __restrict__ z0, const T* __restrict__ z1,
                                                          const T* __restrict__ z2, int B, int K,
                                                          float* __restrict__ row_loss, T* __restrict__ d0,
                                                          T* __restrict__ d1, T* __restrict__ d2) {
    __shared__ float lds[4];
    const int row = blockIdx.x;
    const T* z[3] = {z0 + (size_t)row * K, z1 + (size_t)row * K, z2 + (size_t)row * K};
    T* d[3] = {d0 ? d0 + (size_t)row * K : nullptr, d1 ? d1 + (size_t)row * K : nullptr,
               d2 ? d2 + (size_t)row * K : nullptr};
    const float c = 1.0f / (3.0f * (float)B);

    // pass 1: log-softmax normalisers  lse_i = max_i + log(sum exp(z - max_i))
    float mx[3], lse[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float m = -INFINITY;
        for (int k = threadIdx.x; k < K; k += kBlock) m = fmaxf(m, to_float(z[i][k]));
        mx[i] = block_max(m, lds);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float s = 0.f;
        for (int k = threadIdx.x; k < K; k += kBlock) s += expf(to_float(z[i][k]) - mx[i]);
        lse[i] = mx[i] + logf(block_sum(s, lds));
    }

    // pass 2: loss terms and S_i = sum_k p_ik g_ik / c
    float loss = 0.f, S[3] = {0.f, 0.f, 0.f};
    for (int k = threadIdx.x; k < K; k += kBlock) {
        float lp[3], p[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            lp[i] = to_float(z[i][k]) - lse[i];
            p[i] = expf(lp[i]);
        }
        const float m = (p[0] + p[1] + p[2]) * (1.0f / 3.0f);
        const bool inside = m >= 1e-7f && m <= 1.0f;
        const float lm = logf(fminf(fmaxf(m, 1e-7f), 1.0f));
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (p[i] > 0.f) {
                const float t = lp[i] - lm;
                loss = fmaf(p[i], t, loss);
                S[i] = fmaf(p[i], inside ? t : t + 1.0f, S[i]);
            }
        }
    }
    loss = block_sum(loss, lds);
    if (threadIdx.x == 0) row_loss[row] = loss;
    if (!d[0]) return;  // (uniform) forward only
#pragma unroll
    for (int i = 0; i < 3; ++i) S[i] = block_sum(S[i], lds);

    // pass 3: dL/dz_ik = c p_ik (g'_ik - S_i)
    for (int k = threadIdx.x; k < K; k += kBlock) {
        float lp[3], p[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            lp[i] = to_float(z[i][k]) - lse[i];
            p[i] = expf(lp[i]);
        }
        const float m = (p[0] + p[1] + p[2]) * (1.0f / 3.0f);
        const bool inside = m >= 1e-7f && m <= 1.0f;
        const float lm = logf(fminf(fmaxf(m, 1e-7f), 1.0f));
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float g = 0.f;
            if (p[i] > 0.f) {
                const float t = lp[i] - lm;
                g = c * p[i] * ((inside ? t : t + 1.0f) - S[i]);
            }
            d[i][k] = from_float<T>(g);
        }
    }
}

// loss = sum(rows) / (3B), rows added in index order by one workgroup (deterministic)
__global__ __launch_bounds__(kBlock) void jsd_finish_kernel(const float* __restrict__ row_loss, int B,
                                                            float* __restrict__ loss) {
    __shared__ double part[kBlock];
    double s = 0.0;
    for (int r = threadIdx.x; r < B; r += kBlock) s += (double)row_loss[r];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = kBlock / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(part[0] / (3.0 * (double)B));
}

}  // namespace

extern "C" {

size_t cnsn_jsd_workspace_bytes(int B) { return B > 0 ? (size_t)B * sizeof(float) : 0; }

int cnsn_jsd(const void* logits_clean, const void* logits_aug1, const void* logits_aug2, int dtype, int B, int K,
             float* loss, void* d_clean, void* d_aug1, void* d_aug2, void* workspace, size_t workspace_bytes,
             void* stream_) {
    if (!logits_clean || !logits_aug1 || !logits_aug2 || !loss || !workspace) return CNSN_E_NULL;
    if (dtype != CNSN_F32 && dtype != CNSN_BF16 && dtype != CNSN_F16) return CNSN_E_DTYPE;
    if (B <= 0 || K <= 0) return CNSN_E_SHAPE;
    const bool want = d_clean || d_aug1 || d_aug2;
    if (want && !(d_clean && d_aug1 && d_aug2)) return CNSN_E_NULL;  // all three gradients or none
    if (workspace_bytes < cnsn_jsd_workspace_bytes(B)) return CNSN_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    float* rows = (float*)workspace;
    auto run = [&](auto tt) {
        using T = typename decltype(tt)::type;
        jsd_rows_kernel<T><<<B, kBlock, 0, stream>>>((const T*)logits_clean, (const T*)logits_aug1, (const T*)logits_aug2,
                                                    B, K, rows, (T*)d_clean, (T*)d_aug1, (T*)d_aug2);
    };
    if (dtype == CNSN_F32)
        run(TypeTag<float>{});
    else if (dtype == CNSN_BF16)
        run(TypeTag<bf16_t>{});
    else
        run(TypeTag<_Float16>{});
    jsd_finish_kernel<<<1, kBlock, 0, stream>>>(rows, B, loss);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CNSN_OK : (int)e;
}

}  // extern "C"
