// Timing harness for k_dense_f32: compiles lantern_amd/csrc/bruteforce.hip into this one program and times the
// 1024 x 65536 x 768 launch (plain and with the fused top-k epilogue at a radius nothing passes) with HIP events.  Not part of
// the library.  (The r3 diagnostics that led to the pipelined kernel -- the old and the new loop without their loads / LDS
// traffic / barriers -- were built from this harness with macros that are gone again: profiles/r03_dense_variants.md has
// their numbers.)
#include "../../lantern_amd/csrc/bruteforce.hip"
#include <cstdio>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while(0)

int main()
{
    const uint32_t nq = 1024, nb = 65536, dim = 768, kk = 40, cap = 1024;
    std::vector<float> hq((size_t)nq * dim), hb((size_t)nb * dim);
    std::mt19937 g(5);
    std::normal_distribution<float> nd;
    for(auto &v : hq) v = nd(g);
    for(auto &v : hb) v = nd(g);
    float *Q, *B, *qn, *bn, *out;
    uint64_t *best, *cand;
    uint32_t *cnt;
    CK(hipMalloc(&Q, hq.size() * 4)); CK(hipMalloc(&B, hb.size() * 4)); CK(hipMalloc(&qn, nq * 4)); CK(hipMalloc(&bn, nb * 4));
    CK(hipMalloc(&out, (size_t)nq * nb * 4)); CK(hipMalloc(&best, (size_t)nq * kk * 8)); CK(hipMalloc(&cand, (size_t)nq * cap * 8)); CK(hipMalloc(&cnt, nq * 4));
    CK(hipMemcpy(Q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(best, 0, (size_t)nq * kk * 8));  // ordered distance 0 = below every real distance: nothing is appended
    CK(hipMemset(cnt, 0, nq * 4));
    CK(lgpu::launch_row_norms((const uint4 *)Q, nq, dim / 4, qn, 0)); CK(lgpu::launch_row_norms((const uint4 *)B, nb, dim / 4, bn, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for(int fused = 0; fused < 2; ++fused) {
        auto go = [&]() {
            return fused ? lgpu::launch_dense_topk(lgpu::M_COS, (const uint4 *)Q, nq, (const uint4 *)B, nb, dim / 4, qn, bn, best, kk, cand, cnt, cap, 0, 0)
                         : lgpu::launch_dense(lgpu::M_COS, (const uint4 *)Q, nq, (const uint4 *)B, nb, dim / 4, qn, bn, out, nb, 0);
        };
        for(int i = 0; i < 5; ++i) CK(go());
        CK(hipDeviceSynchronize());
        const int reps = 40;
        CK(hipEventRecord(e0, 0));
        for(int i = 0; i < reps; ++i) CK(go());
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, tf = 2.0 * nq * nb * dim / (us * 1e-6) / 1e12;
        double sum = 0;
        if(!fused) {
            std::vector<float> h(4096);
            CK(hipMemcpy(h.data(), out + (size_t)517 * nb + 4096, 4096 * 4, hipMemcpyDeviceToHost));
            for(float v : h) sum += v;
        }
        printf("{\"fused\": %d, \"us_per_launch\": %.1f, \"tflops\": %.1f, \"frac_of_157.3\": %.3f, \"check\": %.6f}\n", fused, us, tf,
               tf / 157.3, sum);
    }
    return 0;
}
