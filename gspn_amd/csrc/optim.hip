// Adam over ONE flat fp32 buffer (parameters, gradients and both moments are flat and contiguous: gspn_amd/parallel.py keeps the MLP
// parameters as views of one buffer and their gradients in FlatGradBucket).  The reference trains with tf.train.AdamOptimizer
// (train_*.py); the update rule below is torch.optim.Adam's (eps added to the bias-corrected sqrt(v)):
//   m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// One launch for the whole model (~0.25 M parameters: 3 us) instead of a multi-tensor launch per ~50 tensors.
#include "common.h"

__global__ void adam_flat_kernel(long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float weight_decay, float grad_scale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * grad_scale;
        const float pi = p[i];
        if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
        const float mi = fmaf(1.f - b1, gi - m[i], m[i]);                  // lerp, as torch does
        const float vi = fmaf(1.f - b2, gi * gi, b2 * v[i]);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}
// step >= 1 is the number of this update (bias corrections 1 - b^step are computed on the host in double)
// grad_scale multiplies every gradient first (1/world_size after a SUM all-reduce: saves the separate division kernel)
extern "C" int gspn_adam_flat(long n, float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps, float weight_decay,
                              float grad_scale, long step, void* stream) {
    if (n < 0 || step < 1) return GSPN_ERR_ARG;
    if (n == 0) return 0;
    if (!p || !g || !m || !v) return GSPN_ERR_ARG;
    const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
    hipLaunchKernelGGL(adam_flat_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, n, p, g, m, v, lr, b1, b2, eps, (float)bc1,
                       (float)sqrt(bc2), weight_decay, grad_scale);
    return gspn_launch_status();
}

// The same update with the step number kept ON THE DEVICE (r06): state[0] = number of updates done so far, state[1] = scratch (0 between launches).
// A launch whose arguments never change can be a node of a captured hipGraph -- the optimiser then replays with the step instead of being enqueued
// behind it by the host (an ~8 us bubble per step).  Every workgroup reads state[0] when it starts; a workgroup's last act is a ticket on state[1],
// and the workgroup that draws the last ticket -- necessarily after every other one has read state[0] (its value has been USED by then) -- publishes
// step + 1 and clears the tickets.  Relaxed device-scope atomics: nothing but these two words is communicated, and an acquire / release at agent scope
// would write back and invalidate the L2 once per workgroup (measured: +30 us on a 4 us kernel); the next launch sees the words through the kernel boundary.
// Bias corrections in double on the device (pow of a float base: same values as the host's computation in gspn_adam_flat).
__global__ void adam_flat_dev_kernel(long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                     float lr, float b1, float b2, float eps, float weight_decay, float grad_scale, unsigned long long* __restrict__ state) {
    const unsigned long long step = __atomic_load_n(&state[0], __ATOMIC_RELAXED) + 1ull;
    const float bc1 = (float)(1.0 - pow((double)b1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * grad_scale;
        const float pi = p[i];
        if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
        const float mi = fmaf(1.f - b1, gi - m[i], m[i]);
        const float vi = fmaf(1.f - b2, gi * gi, b2 * v[i]);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long ticket = __hip_atomic_fetch_add(&state[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == (unsigned long long)gridDim.x - 1ull) {
            __hip_atomic_store(&state[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&state[0], step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
extern "C" int gspn_adam_flat_dev(long n, float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps, float weight_decay,
                                  float grad_scale, unsigned long long* state, void* stream) {
    if (n < 0 || !state) return GSPN_ERR_ARG;
    if (n == 0) return 0;
    if (!p || !g || !m || !v || ((uintptr_t)state & 7)) return GSPN_ERR_ARG;
    hipLaunchKernelGGL(adam_flat_dev_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, n, p, g, m, v, lr, b1, b2, eps, weight_decay,
                       grad_scale, state);
    return gspn_launch_status();
}

// <a, b> over n floats, deterministic: DOT_BLOCKS workgroups leave one partial each (per-thread sums in grid-stride order, wave shuffle tree,
// one LDS hop), a second tiny launch adds the partials in index order, in double.  The loss of a training step as a product with a constant
// tensor (bench.py) -- two streams at HBM rate instead of a library reduction.  work: DOT_BLOCKS floats.
#ifndef DOT_BLOCKS
#define DOT_BLOCKS 1024
#endif
__global__ __launch_bounds__(256) void dot_partial_kernel(long n4, long n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part) {
    __shared__ float sw[4];
    float s = 0.f;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x = a4[i], y = b4[i];
        s += x.x * y.x; s += x.y * y.y; s += x.z * y.z; s += x.w * y.w;
    }
    if (blockIdx.x == 0) for (long i = 4 * n4 + threadIdx.x; i < n; i += 256) s += a[i] * b[i];          // tail
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) s += __shfl_xor(s, sft, 64);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ __launch_bounds__(64) void dot_final_kernel(int nparts, const float* __restrict__ part, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += (double)part[i];
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) s += __shfl_xor(s, sft, 64);
    if (threadIdx.x == 0) out[0] = (float)s;
}
extern "C" long gspn_dot_work_floats(void) { return DOT_BLOCKS; }
extern "C" int gspn_dot(long n, const float* a, const float* b, float* work, float* out, void* stream) {
    if (n < 0 || !out || !work || (n > 0 && (!a || !b))) return GSPN_ERR_ARG;
    const bool vec = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
    const long n4 = vec ? n / 4 : 0;
    long nb = (n4 + 255) / 256;
    nb = nb < 1 ? 1 : (nb > DOT_BLOCKS ? DOT_BLOCKS : nb);
    hipLaunchKernelGGL(dot_partial_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, n4, n, a, b, work);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int)nb, work, out);
    return gspn_launch_status();
}

