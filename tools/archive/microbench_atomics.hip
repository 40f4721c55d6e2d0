#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
// each block adds its own values into the SAME array of `span` floats (split-K style reduction)
__global__ void k_same(float* dst, int span, int reps) {
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < span; i += blockDim.x) atomicAdd(&dst[i], 1.0f);
}
__global__ void k_same_unsafe(float* dst, int span, int reps) {
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < span; i += blockDim.x) unsafeAtomicAdd(&dst[i], 1.0f);
}
__global__ void k_rand(float* dst, int span, int per_thread) {
    unsigned s = blockIdx.x * 9781u + threadIdx.x * 6271u + 1u;
    for (int r = 0; r < per_thread; ++r) { s = s * 1664525u + 1013904223u; atomicAdd(&dst[(s >> 8) % span], 1.0f); }
}
__global__ void k_int(int* dst, int span, int per_thread) {
    unsigned s = blockIdx.x * 9781u + threadIdx.x * 6271u + 1u;
    for (int r = 0; r < per_thread; ++r) { s = s * 1664525u + 1013904223u; atomicAdd(&dst[(s >> 8) % span], 1); }
}
int main() {
    float* d; hipMalloc(&d, 64 << 20); hipMemset(d, 0, 64 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto launch, double natom) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-40s %8.3f ms  %8.2f G atomics/s\n", name, ms, natom / ms / 1e6);
    };
    for (int span : {32768, 131072}) for (int G : {32, 128, 512}) {
        char nm[64]; snprintf(nm, 64, "same span=%d G=%d", span, G);
        run(nm, [&] { hipLaunchKernelGGL(k_same, dim3(G), dim3(256), 0, 0, d, span, 4); }, (double)G * span * 4);
        snprintf(nm, 64, "same-unsafe span=%d G=%d", span, G);
        run(nm, [&] { hipLaunchKernelGGL(k_same_unsafe, dim3(G), dim3(256), 0, 0, d, span, 4); }, (double)G * span * 4);
    }
    for (int span : {1 << 16, 1 << 20, 1 << 24}) {
        char nm[64]; snprintf(nm, 64, "rand f32 span=%d", span);
        run(nm, [&] { hipLaunchKernelGGL(k_rand, dim3(2048), dim3(256), 0, 0, d, span, 64); }, 2048.0 * 256 * 64);
        snprintf(nm, 64, "rand i32 span=%d", span);
        run(nm, [&] { hipLaunchKernelGGL(k_int, dim3(2048), dim3(256), 0, 0, (int*)d, span, 64); }, 2048.0 * 256 * 64);
    }
    return 0;
}
