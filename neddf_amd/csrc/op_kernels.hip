// op_kernels.hip -- the reference's (value, Jacobian) layer ops as stand-alone
// elementwise kernels (neddf/nn_module/with_grad/*.py forward halves,
// neddf/nn_module/{positional_encoding,tanh_exp}.py).  The renderer never calls
// these -- the same device functions (device_math.h) run inside the fused field
// kernels' epilogues -- they exist so that each op has a drop-in counterpart and
// a unit-level parity test on the GPU.
#include "kernels.h"
#include "device_math.h"

namespace neddf {

// kind: 0 ReLU, 1 LeakyReLU, 2 tanhExp, 3 Softplus, 4 Sigmoid.  x [N,C], J [N,3,C] (J may be NULL: value only)
__global__ void op_activation_kernel(int kind, const float *x, const float *J, int64_t N, int C, float *y, float *G)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    int64_t n = i / C;
    int c = (int)(i - n * C);
    float yy, dy;
    float v = x[i];
    switch (kind) {
    case 0: act_grad<0>(v, yy, dy); break;
    case 1: act_grad<1>(v, yy, dy); break;
    case 2: act_grad<2>(v, yy, dy); break;
    case 3: softplus_grad(v, yy, dy); break;
    default: sigmoid_grad(v, yy, dy); break;
    }
    if (!J) {       // plain activations: F.relu / F.leaky_relu / tanhExp.apply
        if (kind == 0) yy = act_val<0>(v);
        else if (kind == 1) yy = act_val<1>(v);
        else if (kind == 2) yy = act_val<2>(v);
        y[i] = yy;
        return;
    }
    y[i] = yy;
#pragma unroll
    for (int k = 0; k < 3; ++k) G[(n * 3 + k) * C + c] = dy * J[(n * 3 + k) * C + c];
}

// PositionalEncoding(.GradLayer).forward: x [N,3], J [N,3,3] or NULL, scale [N,3E] or NULL -> y [N,6E], G [N,3,6E]
__global__ void op_pe_kernel(const float *x, const float *J, const float *scale, int64_t N, int E, float *y, float *G)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int C = 3 * E;
    if (i >= N * C) return;
    int64_t n = i / C;
    int c = (int)(i - n * C), e = c / 3, d = c - 3 * e;
    float f = (float)(1 << e);
    float s = scale ? scale[i] : 1.0f;
    float sn, cs;
    sincosf(f * x[n * 3 + d], &sn, &cs);
    y[n * 2 * C + c] = s * sn;
    y[n * 2 * C + C + c] = s * cs;
    if (J && G)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float sG = f * s * J[(n * 3 + k) * 3 + d];          // with_grad/positional_encoding.py:70-79
            G[(n * 3 + k) * 2 * C + c] = sG * cs;
            G[(n * 3 + k) * 2 * C + C + c] = -sG * sn;
        }
}

// Sampling.get_pe_weights sampling.py:55-71
__global__ void op_pe_weights_kernel(const float *var, int64_t N, int E, float *w)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int C = 3 * E;
    if (i >= N * C) return;
    int64_t n = i / C;
    int c = (int)(i - n * C), e = c / 3, d = c - 3 * e;
    float f = (float)(1 << e);
    w[i] = expf(-0.5f * (f * f) * var[n * 3 + d]);
}

void launch_op_activation(int kind, const float *x, const float *J, int64_t N, int C, float *y, float *G, hipStream_t s)
{
    int64_t t = N * C;
    if (t > 0) hipLaunchKernelGGL(op_activation_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, kind, x, J, N, C, y, G);
}
void launch_op_pe(const float *x, const float *J, const float *scale, int64_t N, int E, float *y, float *G, hipStream_t s)
{
    int64_t t = N * 3 * E;
    if (t > 0) hipLaunchKernelGGL(op_pe_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, x, J, scale, N, E, y, G);
}
void launch_op_pe_weights(const float *var, int64_t N, int E, float *w, hipStream_t s)
{
    int64_t t = N * 3 * E;
    if (t > 0) hipLaunchKernelGGL(op_pe_weights_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, var, N, E, w);
}

}  // namespace neddf
