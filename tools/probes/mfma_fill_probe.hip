// Probe: how many ordinary vector instructions of the SAME wave hide behind a v_mfma_f32_16x16x4_f32 (32 cycles of the matrix
// pipe)?  Compiler-scheduled (builtins + sched_group_barrier), one wave per SIMD.  KIND 0: v_fma_f32, 1: v_cndmask (select),
// 2: v_add_f32; the fillers run on 16 independent registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NF, int KIND>
__global__ void __launch_bounds__(256, 1) probe(int nm, float seed, unsigned long long* cycles, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = seed + k + lane;
    float a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = seed * (k + 1) + lane; b[k] = seed * 0.5f + lane * (k + 2); }
    const float m = 1.0000001f + seed * 1e-8f, ad = seed * 1e-9f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < nm; i += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], c[k], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float& x = v[(k * NF + f) & 15];
                if constexpr (KIND == 0) x = __builtin_fmaf(x, m, ad);
                else if constexpr (KIND == 1) x = (x > ad) ? x : m;
                else x = x + ad;
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (NF > 0) __builtin_amdgcn_sched_group_barrier(0x002, KIND == 1 ? 2 * NF : NF, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc_out = c[0][0] + c[1][1] + c[2][2] + c[3][3];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc_out += v[k];
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
    if (acc_out == 12345.678f) sink[threadIdx.x] = acc_out;
}

template <int NF, int KIND> void run(int nm) {
    unsigned long long* d_c; float* d_s;
    const int nb = 256;
    (void)hipMalloc(&d_c, nb * 4 * 8); (void)hipMalloc(&d_s, 4096);
    std::vector<unsigned long long> h(nb * 4);
    for (int rep = 0; rep < 3; ++rep) probe<NF, KIND><<<nb, 256>>>(nm, 1.0f, d_c, d_s);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_c, nb * 4 * 8, hipMemcpyDeviceToHost);
    std::vector<double> m(h.begin(), h.end());
    std::sort(m.begin(), m.end());
    printf("kind %d: f32 16x16x4 + %d fillers each: %.1f cycles per matrix instruction\n", KIND, NF, m[m.size() / 2] / nm);
    (void)hipFree(d_c); (void)hipFree(d_s);
}

int main() {
    run<0, 0>(8192); run<1, 0>(8192); run<2, 0>(8192); run<3, 0>(8192); run<4, 0>(8192); run<5, 0>(8192); run<6, 0>(8192); run<8, 0>(8192); run<12, 0>(8192);
    run<2, 2>(8192); run<4, 2>(8192); run<6, 2>(8192); run<8, 2>(8192);
    run<2, 1>(8192); run<4, 1>(8192);
    return 0;
}
