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
