// Stand-alone check + timing of the layer weight-gradient contraction (geomae_amd/csrc/dw_device.h), outside the library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I geomae_amd/csrc tools/dw_bench.hip -o tools/dw_bench && tools/dw_bench [n] [G] [reps]
// One SST layer's four jobs (QK, W1 with the LayerNorm-1 affine folded, W2, VO with the LayerNorm-2 affine on V's operand)
// on random tile-blocked bf16 operands; results against a double-precision host contraction at n = 4013 and 2x dead rows;
// then launches back to back at the requested size (each carrying the reduction of the one before) under HIP events.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dw_device.h"

using namespace geomae;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static unsigned short f2bf(float f) { unsigned int u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned int u = (unsigned int)h << 16; float f; memcpy(&f, &u, 4); return f; }

struct Tensor {                      // tile-blocked [n16][ld/16][16][16] bf16
    int n, ld;
    std::vector<unsigned short> h;
    unsigned short* d = nullptr;
    float at(int t, int c) const { return bf2f(h[((size_t)(t >> 4) * (ld >> 4) + (c >> 4)) * 256 + (t & 15) * 16 + (c & 15)]); }
    void init(int n_, int ld_, unsigned seed, float pad_value) {
        n = n_; ld = ld_;
        const int n16 = (n + 15) / 16;
        h.assign((size_t)n16 * ld * 16, 0);
        unsigned s = seed * 2654435761u + 12345u;
        for (int t = 0; t < n16 * 16; ++t)
            for (int c = 0; c < ld; ++c) {
                s = s * 1664525u + 1013904223u;
                float v = ((int)((s >> 9) & 0xffff) - 32768) / 32768.0f;
                if (t >= n) v = pad_value;                       // rows past n: must not matter (NaN)
                h[((size_t)(t >> 4) * (ld >> 4) + (c >> 4)) * 256 + (t & 15) * 16 + (c & 15)] = f2bf(v);
            }
        CK(hipMalloc(&d, h.size() * 2));
        CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    void poison_rows(int upto) {     // rows [0, upto): NaN (the dead rows of a decoder's top layer were never written)
        for (int t = 0; t < upto; ++t)
            for (int c = 0; c < ld; ++c) h[((size_t)(t >> 4) * (ld >> 4) + (c >> 4)) * 256 + (t & 15) * 16 + (c & 15)] = 0x7fc0;
        CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    void zero_rows(int upto) {
        for (int t = 0; t < upto; ++t)
            for (int c = 0; c < ld; ++c) h[((size_t)(t >> 4) * (ld >> 4) + (c >> 4)) * 256 + (t & 15) * 16 + (c & 15)] = 0;
        CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
};

struct Layer {
    int n;
    Tensor dqkv, xp, xh2, du, attn, dhp, xh1, dv, h;
    float *g_wqkv, *g_bqkv, *g_wo, *g_bo, *g_w1, *g_b1, *g_w2, *g_b2;   // device gradients
    float *gam1, *bet1, *gam2, *bet2;
    std::vector<float> hg1, hb1, hg2, hb2;
    void init(int n_, float pad, unsigned so = 0) {
        n = n_;
        dqkv.init(n, 384, 1 + so, pad); xp.init(n, 128, 2 + so, pad); xh2.init(n, 128, 3 + so, pad); du.init(n, 128, 4 + so, pad);
        attn.init(n, 128, 5 + so, pad); dhp.init(n, 256, 6 + so, pad); xh1.init(n, 128, 7 + so, pad); dv.init(n, 128, 8 + so, pad); h.init(n, 256, 9 + so, pad);
        auto zalloc = [](float** p, size_t cnt) { CK(hipMalloc(p, cnt * 4)); CK(hipMemset(*p, 0, cnt * 4)); };
        zalloc(&g_wqkv, 384 * 128); zalloc(&g_bqkv, 384); zalloc(&g_wo, 128 * 128); zalloc(&g_bo, 128);
        zalloc(&g_w1, 256 * 128); zalloc(&g_b1, 256); zalloc(&g_w2, 128 * 256); zalloc(&g_b2, 128);
        hg1.resize(128); hb1.resize(128); hg2.resize(128); hb2.resize(128);
        for (int i = 0; i < 128; ++i) { hg1[i] = 0.8f + 0.003f * i; hb1[i] = 0.05f * ((i % 7) - 3); hg2[i] = 1.1f - 0.002f * i; hb2[i] = 0.03f * ((i % 5) - 2); }
        auto up = [](float** p, const std::vector<float>& v) { CK(hipMalloc(p, v.size() * 4)); CK(hipMemcpy(*p, v.data(), v.size() * 4, hipMemcpyHostToDevice)); };
        up(&gam1, hg1); up(&bet1, hb1); up(&gam2, hg2); up(&bet2, hb2);
    }
    void jobs(DlJob* J, int dead) const {
        memset(J, 0, 4 * sizeof(DlJob));
        J[0].kind = kDlTall; J[0].tok_begin = 0;                                  // QK
        J[0].s[0] = {dqkv.d, 24, 0, 16}; J[0].s[1] = {xp.d, 8, 0, 8};
        J[0].out[0] = {g_wqkv, g_bqkv, nullptr, nullptr, 128, 0};
        J[1].kind = kDlTall; J[1].tok_begin = dead;                               // W1
        J[1].s[0] = {dhp.d, 16, 0, 16}; J[1].s[1] = {xh1.d, 8, 0, 8};
        J[1].out[0] = {g_w1, g_b1, gam1, bet1, 128, 0};
        J[2].kind = kDlWide; J[2].tok_begin = dead;                               // W2
        J[2].s[0] = {dv.d, 8, 0, 8}; J[2].s[1] = {h.d, 16, 0, 16};
        J[2].out[0] = {g_w2, g_b2, nullptr, nullptr, 256, 0};
        J[3].kind = kDlDual; J[3].tok_begin = 0;                                  // VO
        J[3].s[0] = {dqkv.d, 24, 16, 8}; J[3].s[1] = {xh2.d, 8, 0, 8};
        J[3].s[2] = {du.d, 8, 0, 8}; J[3].s[3] = {attn.d, 8, 0, 8};
        J[3].out[0] = {g_wqkv + 256 * 128, g_bqkv + 256, gam2, bet2, 128, 0};
        J[3].out[1] = {g_wo, g_bo, nullptr, nullptr, 128, 0};
    }
};

static DlArgs args_of(const std::vector<Layer*>& Ls, int G, int dead, float* partial) {
    DlArgs A;
    memset(&A, 0, sizeof(A));
    A.njobs = 4 * (int)Ls.size(); A.n = Ls[0]->n; A.G = G; A.partial = partial;
    for (size_t l = 0; l < Ls.size(); ++l) Ls[l]->jobs(&A.job[4 * l], dead);
    return A;
}

static DlReduce reduce_of(const DlArgs& A) {
    DlReduce R;
    R.partial = A.partial; R.njobs = A.njobs; R.G = A.G;
    for (int j = 0; j < A.njobs; ++j) { R.job[j].kind = A.job[j].kind; R.job[j].out[0] = A.job[j].out[0]; R.job[j].out[1] = A.job[j].out[1]; }
    return R;
}

// host reference: C[i][j] = sum_t A[t][a0 + i] * (sc[j] * B[t][b0 + j] + sh[j]), bias[i] = sum_t A[t][a0 + i]
static double check(const char* name, const Tensor& A, int a0, int rows, const Tensor& B, int b0, int cols, const float* sc,
                    const float* sh, int t0, const float* dC, int ldc, const float* dbias) {
    std::vector<double> C((size_t)rows * cols, 0.0), bias(rows, 0.0);
    std::vector<float> arow(rows), brow(cols);
    for (int t = t0; t < A.n; ++t) {
        for (int i = 0; i < rows; ++i) arow[i] = A.at(t, a0 + i);
        for (int j = 0; j < cols; ++j) { const float b = B.at(t, b0 + j); brow[j] = sc ? sc[j] * b + sh[j] : b; }
        for (int i = 0; i < rows; ++i) {
            bias[i] += arow[i];
            double* c = &C[(size_t)i * cols];
            const double a = arow[i];
            for (int j = 0; j < cols; ++j) c[j] += a * brow[j];
        }
    }
    std::vector<float> got((size_t)rows * ldc), gb(rows);
    CK(hipMemcpy(got.data(), dC, got.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gb.data(), dbias, rows * 4, hipMemcpyDeviceToHost));
    if (!sc && dC && rows == 128 && cols == 128 && name[0] == 'd') for (auto& b : bias) b = 0;      // (SPLIT: no bias output)
    double emax = 0, cmax = 0, bmax = 0, bref = 0;
    int bad_nan = 0;
    for (int i = 0; i < rows; ++i) {
        for (int j = 0; j < cols; ++j) {
            const double g = got[(size_t)i * ldc + j], w = C[(size_t)i * cols + j];
            if (!(g == g)) ++bad_nan;
            emax = fmax(emax, fabs(g - w)); cmax = fmax(cmax, fabs(w));
        }
        bmax = fmax(bmax, fabs(gb[i] - bias[i])); bref = fmax(bref, fabs(bias[i]));
    }
    printf("  %-3s max|err| %.3e of max|C| %.3e (rel %.2e)   bias err %.3e of %.3e   NaNs %d\n", name, emax, cmax, emax / cmax, bmax, bref, bad_nan);
    return fmax(emax / cmax, bmax / fmax(bref, 1e-30)) + (bad_nan ? 1.0 : 0.0);
}

static void launch(const DlArgs& A, hipStream_t s) {
    hipLaunchKernelGGL(dw_layer_kernel, dim3(A.njobs * A.G), dim3(kDlThreads), 0, s, A);
}
static void reduce(const DlArgs& A, hipStream_t s) {
    const DlReduce R = reduce_of(A);
    hipLaunchKernelGGL(dw_layer_reduce_kernel, dim3((A.njobs * kDlTileSlots + 255) / 256), dim3(256), 0, s, R);
}

int main(int argc, char** argv) {
    const int n_time = argc > 1 ? atoi(argv[1]) : 22016;
    const int G = argc > 2 ? atoi(argv[2]) : 24;
    const int reps = argc > 3 ? atoi(argv[3]) : 50;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    double worst = 0;
    float* partial;
    CK(hipMalloc(&partial, (size_t)kDlMaxJobs * 16 * kDlPartialFloats * 4));      // (also >= 96 chunks of one SPLIT job)
    for (int variant = 0; variant < 2; ++variant) {
        const int n = variant == 0 ? 4013 : 2500, dead = variant == 0 ? 0 : 640;
        Layer L0, L1;
        L0.init(n, NAN, 0); L1.init(n, NAN, 100);
        for (Layer* L : {&L0, &L1})
            if (dead) { L->xh1.poison_rows(dead); L->dhp.zero_rows(dead); L->dv.zero_rows(dead); L->h.poison_rows(dead); }
        const int Gv = variant == 0 ? 7 : 16;
        const DlArgs A = args_of({&L0, &L1}, Gv, dead, partial);         // two layers = 8 jobs in one launch
        for (int pass = 0; pass < 3; ++pass) { launch(A, s); reduce(A, s); }
        CK(hipStreamSynchronize(s));
        printf("n = %d, G = %d, dead rows %d, 2 layers per launch: gradients after THREE accumulations / 3 vs host\n", n, Gv, dead);
        auto third = [&](float* p, size_t cnt) {
            std::vector<float> v(cnt);
            CK(hipMemcpy(v.data(), p, cnt * 4, hipMemcpyDeviceToHost));
            for (auto& x : v) x /= 3.0f;
            CK(hipMemcpy(p, v.data(), cnt * 4, hipMemcpyHostToDevice));
        };
        for (Layer* Lp : {&L0, &L1}) {
            Layer& L = *Lp;
            third(L.g_wqkv, 384 * 128); third(L.g_bqkv, 384); third(L.g_wo, 128 * 128); third(L.g_bo, 128);
            third(L.g_w1, 256 * 128); third(L.g_b1, 256); third(L.g_w2, 128 * 256); third(L.g_b2, 128);
            worst = fmax(worst, check("QK", L.dqkv, 0, 256, L.xp, 0, 128, nullptr, nullptr, 0, L.g_wqkv, 128, L.g_bqkv));
            worst = fmax(worst, check("V", L.dqkv, 256, 128, L.xh2, 0, 128, L.hg2.data(), L.hb2.data(), 0, L.g_wqkv + 256 * 128, 128, L.g_bqkv + 256));
            worst = fmax(worst, check("O", L.du, 0, 128, L.attn, 0, 128, nullptr, nullptr, 0, L.g_wo, 128, L.g_bo));
            worst = fmax(worst, check("W1", L.dhp, 0, 256, L.xh1, 0, 128, L.hg1.data(), L.hb1.data(), dead, L.g_w1, 128, L.g_b1));
            worst = fmax(worst, check("W2", L.dv, 0, 128, L.h, 0, 256, nullptr, nullptr, dead, L.g_w2, 256, L.g_b2));
        }
    }
    // ---- the SPLIT kind: one [128 x 128] product over many tokens (the VFE's layer-1 weight gradient)
    {
        const int n = 5003;
        Tensor A, B;
        A.init(n, 128, 41, NAN); B.init(n, 128, 42, NAN);
        float *gC, *gb;
        CK(hipMalloc(&gC, 128 * 128 * 4)); CK(hipMemset(gC, 0, 128 * 128 * 4)); CK(hipMalloc(&gb, 128 * 4)); CK(hipMemset(gb, 0, 128 * 4));
        DlArgs S;
        memset(&S, 0, sizeof(S));
        S.njobs = 1; S.n = n; S.G = 13; S.partial = partial;
        S.job[0].kind = kDlSplit;
        S.job[0].s[0] = {A.d, 8, 0, 8}; S.job[0].s[1] = {B.d, 8, 0, 8}; S.job[0].s[2] = {A.d, 8, 16, 8}; S.job[0].s[3] = {B.d, 8, 16, 8};
        S.job[0].out[0] = {gC, nullptr, nullptr, nullptr, 128, 0};
        launch(S, s); reduce(S, s);
        CK(hipStreamSynchronize(s));
        printf("SPLIT job, n = %d, G = %d:\n", n, S.G);
        worst = fmax(worst, check("dW1", A, 0, 128, B, 0, 128, nullptr, nullptr, 0, gC, 128, gb) - 0.0);
    }
    printf("worst relative error %.3e -> %s\n", worst, worst < 2e-5 ? "OK" : "FAIL");
    for (int n : {106000, 1030000}) {
        Tensor A, B;
        A.init(n, 128, 51, 0.f); B.init(n, 128, 52, 0.f);
        float* gC;
        CK(hipMalloc(&gC, 128 * 128 * 4)); CK(hipMemset(gC, 0, 128 * 128 * 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int Gt : {32, 48, 64, 96}) {
            DlArgs S;
            memset(&S, 0, sizeof(S));
            S.njobs = 1; S.n = n; S.G = Gt; S.partial = partial;
            S.job[0].kind = kDlSplit;
            S.job[0].s[0] = {A.d, 8, 0, 8}; S.job[0].s[1] = {B.d, 8, 0, 8}; S.job[0].s[2] = {A.d, 8, 16, 8}; S.job[0].s[3] = {B.d, 8, 16, 8};
            S.job[0].out[0] = {gC, nullptr, nullptr, nullptr, 128, 0};
            for (int i = 0; i < 3; ++i) { launch(S, s); reduce(S, s); }
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) { launch(S, s); reduce(S, s); }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("SPLIT n = %7d G = %2d: %.1f us with its reduction, operands %.1f MB -> %.2f TB/s\n", n, Gt, ms * 1e3 / reps, 512e-6 * n,
                   512.0 * n / (ms * 1e3 / reps) * 1e-6);
        }
    }

    // ---- timing: `layers` layers per launch + its reduction, back to back
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    for (int n : {n_time, 6592}) {
        std::vector<Layer> Ls(4);
        for (int l = 0; l < 4; ++l) Ls[l].init(n, 0.f, 10 * l);
        for (int layers : {4, 1}) {
            std::vector<Layer*> P;
            for (int l = 0; l < layers; ++l) P.push_back(&Ls[l]);
            for (int Gt : {G, 6, 8, 12, 16, 24}) {
                if (4 * layers * Gt * (size_t)kDlPartialFloats > (size_t)kDlMaxJobs * 16 * kDlPartialFloats) continue;
                const DlArgs A = args_of(P, Gt, 0, partial);
                for (int i = 0; i < 4; ++i) { launch(A, s); reduce(A, s); }
                float ms_all, ms_dw = 0;
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < reps; ++i) { launch(A, s); reduce(A, s); }
                CK(hipEventRecord(e1, s));
                for (int i = 0; i < reps; ++i) launch(A, s);
                CK(hipEventRecord(e2, s));
                CK(hipEventSynchronize(e2));
                CK(hipEventElapsedTime(&ms_all, e0, e1));
                CK(hipEventElapsedTime(&ms_dw, e1, e2));
                const double us = ms_all * 1e3 / reps, usd = ms_dw * 1e3 / reps, alg = 4096.0 * n * layers;
                const double act = 3328.0 * n * layers + 2.0 * 4 * layers * Gt * kDlPartialFloats * 4;
                printf("n = %6d, %d layer(s) per launch, G = %2d (%3d workgroups): %.1f us with its reduction (%.1f without) = %.1f us per layer; "
                       "algorithmic %.2f TB/s, moved %.1f MB -> %.2f TB/s\n", n, layers, Gt, 4 * layers * Gt, us, usd, us / layers,
                       alg / us * 1e-6, act * 1e-6, act / us * 1e-6);
            }
        }
    }
    return worst < 2e-5 ? 0 : 1;
}
