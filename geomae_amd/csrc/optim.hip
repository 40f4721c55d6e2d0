// Optimizer step of the pre-training loop, on the flat parameter / gradient buffers of
// geomae_amd.train.FlatParams.
//
// Reference behaviour (external to its tree: mmcv OptimizerHook + torch.optim.AdamW, configured by
// configs/_base_/schedules/cosine_2x.py:1-17): clip_grad_norm_(max_norm=10, norm_type=2) over all
// parameters, then AdamW(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05) with weight decay switched off
// for parameters whose name contains 'norm'.  There it is ~10 multi-tensor launches over ~330 tensors; on the
// flat buffers it was 15 elementwise ATen kernels (8 passes over the 11 MB buffers each way).  Here:
//   grad_sumsq_kernel : sum of squares of the gradient buffer -> one fp64 word (block partials, fp64 atomics)
//   adamw_kernel      : reads that word, forms the clip coefficient, and does the whole update in ONE pass:
//                       4 streams read (p, g, m, v), 3-4 written (p, m, v, g = 0)  = 32 B per parameter.
// The arithmetic follows torch.optim.AdamW's single-tensor path operation by operation in fp32
// (p *= 1 - lr*wd ; m = lerp(m, g, 1-b1) ; v = v*b2 + (1-b2) g g ; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)),
// compiled with -ffp-contract=off, so the result is bit-identical to it given the same clip coefficient.
#include "common.h"
#include "../../include/geomae_hip.h"

namespace geomae {

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ out) {
    __shared__ double red[4];
    double acc = 0.0;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {                    // four independent loads in flight per thread
        const float4 a = g4[i], b = g4[i + stride], c = g4[i + 2 * stride], d = g4[i + 3 * stride];
        acc += (double)(a.x * a.x + a.y * a.y) + (double)(a.z * a.z + a.w * a.w);
        acc += (double)(b.x * b.x + b.y * b.y) + (double)(b.z * b.z + b.w * b.w);
        acc += (double)(c.x * c.x + c.y * c.y) + (double)(c.z * c.z + c.w * c.w);
        acc += (double)(d.x * d.x + d.y * d.y) + (double)(d.z * d.z + d.w * d.w);
    }
    for (; i < n4; i += stride) {
        const float4 v = g4[i];
        acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        acc += (double)(v * v);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

struct AdamwArgs {
    float *p, *g, *m, *v;
    int64_t n, n_no_decay;
    float lr_wd_factor;      // 1 - lr * weight_decay
    float w1;                // 1 - beta1 (lerp weight)
    float b2, one_minus_b2;
    float bc2_sqrt, eps, step_size;
    float max_norm, grad_scale;
    int64_t nd2_start, nd2_end;  // a second no-decay range [start, end) (flat buffers made of two segments), or empty
    double* zero_after;      // optional accumulator word to clear for the NEXT step (not read by this launch)
    const double* sumsq;     // of grad * grad_scale is sumsq * grad_scale^2
    float* gnorm_out;
    int zero_grad;
};

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, bool decay, const AdamwArgs& a, float coef) {
    g = __fmul_rn(g, coef);
    if (decay) p = __fmul_rn(p, a.lr_wd_factor);
    // torch.lerp(m, g, w): w < 0.5 ? m + w * (g - m) : g - (g - m) * (1 - w)
    const float d = __fsub_rn(g, m);
    m = a.w1 < 0.5f ? __fadd_rn(m, __fmul_rn(a.w1, d)) : __fsub_rn(g, __fmul_rn(d, __fsub_rn(1.0f, a.w1)));
    v = __fadd_rn(__fmul_rn(v, a.b2), __fmul_rn(__fmul_rn(a.one_minus_b2, g), g));
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), a.bc2_sqrt), a.eps);
    p = __fadd_rn(p, __fmul_rn(-a.step_size, __fdiv_rn(m, denom)));
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamwArgs a) {
    // clip coefficient of torch.nn.utils.clip_grad_norm_: min(1, max_norm / (norm + 1e-6)), times the
    // data-parallel averaging factor when the caller folded it in here
    float coef = a.grad_scale;
    if (a.sumsq) {
        const float norm = (float)(sqrt(*a.sumsq) * (double)a.grad_scale);
        if (a.gnorm_out && blockIdx.x == 0 && threadIdx.x == 0) *a.gnorm_out = norm;
        if (a.max_norm > 0.f) {
            const float c = a.max_norm / (norm + 1e-6f);
            coef *= c < 1.0f ? c : 1.0f;
        }
    }
    if (a.zero_after && blockIdx.x == 0 && threadIdx.x == 0) *a.zero_after = 0.0;
    auto decays = [&](int64_t e) { return e >= a.n_no_decay && !(e >= a.nd2_start && e < a.nd2_end); };
    const int64_t n4 = a.n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 p = reinterpret_cast<float4*>(a.p)[i], m = reinterpret_cast<float4*>(a.m)[i], v = reinterpret_cast<float4*>(a.v)[i];
        const float4 g = reinterpret_cast<const float4*>(a.g)[i];
        const int64_t e = i << 2;
        adamw_one(p.x, g.x, m.x, v.x, decays(e + 0), a, coef);
        adamw_one(p.y, g.y, m.y, v.y, decays(e + 1), a, coef);
        adamw_one(p.z, g.z, m.z, v.z, decays(e + 2), a, coef);
        adamw_one(p.w, g.w, m.w, v.w, decays(e + 3), a, coef);
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        if (a.zero_grad) reinterpret_cast<float4*>(a.g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const int64_t e = (n4 << 2) + threadIdx.x;
        adamw_one(a.p[e], a.g[e], a.m[e], a.v[e], decays(e), a, coef);
        if (a.zero_grad) a.g[e] = 0.f;
    }
}

}  // namespace geomae

using namespace geomae;

extern "C" int geomae_grad_sumsq(const float* grad, int64_t num_elems, double* sumsq, hipStream_t stream) {
    GEOMAE_REQUIRE(grad && sumsq && num_elems >= 0, "grad_sumsq: bad argument");
    GEOMAE_REQUIRE(((uintptr_t)grad & 15) == 0, "grad_sumsq: buffer must be 16-byte aligned");
    GEOMAE_ZERO(sumsq, sizeof(double), stream);
    if (num_elems == 0) return GEOMAE_OK;
    // one workgroup per CU at most: every workgroup ends in one fp64 atomic on the same word, and 2048 of them
    // serialised in L2 were most of the kernel's 28 us on an 11 MB buffer
    const int64_t wgs = (num_elems / 4 + 2047) / 2048;                 // >= 8 float4 per thread
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3((int)(wgs < 1 ? 1 : (wgs > 256 ? 256 : wgs))), dim3(256), 0, stream, grad,
                       num_elems, sumsq);
    return check_launch("grad_sumsq_kernel");
}

extern "C" int geomae_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t num_elems,
                                 int64_t num_no_decay, float lr, float beta1, float beta2, float eps, float weight_decay,
                                 int64_t step, float max_norm, const double* grad_sumsq, float grad_scale,
                                 int32_t zero_grad, float* grad_norm_out, int64_t no_decay2_start,
                                 int64_t no_decay2_count, double* zero_after, hipStream_t stream) {
    GEOMAE_REQUIRE(params && grads && exp_avg && exp_avg_sq && num_elems >= 0 && num_no_decay >= 0 && num_no_decay <= num_elems,
                   "adamw_step: bad argument");
    GEOMAE_REQUIRE(no_decay2_count >= 0 && (no_decay2_count == 0 || (no_decay2_start >= num_no_decay &&
                   no_decay2_start + no_decay2_count <= num_elems)), "adamw_step: second no-decay range out of bounds");
    GEOMAE_REQUIRE(zero_after != grad_sumsq || !zero_after, "adamw_step: zero_after must not be the word being read");
    GEOMAE_REQUIRE(step >= 1, "adamw_step: step counts from 1");
    GEOMAE_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                   "adamw_step: buffers must be 16-byte aligned");
    GEOMAE_REQUIRE(max_norm <= 0.f || grad_sumsq, "adamw_step: clipping needs grad_sumsq");
    if (num_elems == 0) return GEOMAE_OK;
    AdamwArgs a;
    a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq;
    a.n = num_elems; a.n_no_decay = num_no_decay;
    a.nd2_start = no_decay2_start; a.nd2_end = no_decay2_start + no_decay2_count; a.zero_after = zero_after;
    // python-double scalars rounded to fp32 where torch hands them to an fp32 tensor op
    a.lr_wd_factor = (float)(1.0 - (double)lr * (double)weight_decay);
    a.w1 = (float)(1.0 - (double)beta1);
    a.b2 = beta2;
    a.one_minus_b2 = (float)(1.0 - (double)beta2);
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.eps = eps;
    a.step_size = (float)((double)lr / bc1);
    a.max_norm = max_norm; a.grad_scale = grad_scale;
    a.sumsq = grad_sumsq; a.gnorm_out = grad_norm_out; a.zero_grad = zero_grad;
    hipLaunchKernelGGL(adamw_kernel, dim3(stream_grid(num_elems / 4 + 1, 256)), dim3(256), 0, stream, a);
    return check_launch("adamw_kernel");
}
