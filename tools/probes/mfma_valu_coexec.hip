// Probe: do f32-input matrix instructions (v_mfma_f32_16x16x4_f32) and ordinary vector ALU instructions of ANOTHER wave on
// the same SIMD execute side by side, or do they take turns?  (MI355X_MICROARCH.md: the f32 MFMA runs at exactly the f32
// VECTOR rate, 64 FLOP/clk/SIMD; its co-execution figures were measured with bf16 MFMAs.)
// One 512-thread block per CU: waves w and w + 4 share a SIMD.  Waves 0-3 issue NM matrix instructions (four independent
// accumulators), waves 4-7 issue NV dependent-free v_fma_f32 (eight independent chains).  Each role alone, then both.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND>      // 0: f32 16x16x4 (32 cycles), 1: bf16 16x16x16 (8 cycles)
__global__ void __launch_bounds__(512, 2) probe(int nm, int nv, int run_m, int run_v, float seed, unsigned long long* cycles, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0 = 0, t1 = 0;
    float acc_out = 0.f;
    __syncthreads();
    if (wave < 4) {
        if (run_m) {
            f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const float a = seed + lane, b = seed * 0.5f + lane;
            const s16x4 ah = {(short)lane, 1, 2, 3}, bh = {3, 2, 1, (short)lane};
            t0 = __builtin_readcyclecounter();
            for (int i = 0; i < nm; i += 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (KIND == 0) c[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[k], 0, 0, 0);
                    else c[k] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c[k], 0, 0, 0);
                }
            }
            acc_out = c[0][0] + c[1][1] + c[2][2] + c[3][3];
            t1 = __builtin_readcyclecounter();
        }
    } else if (run_v) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = seed + k + lane;
        const float m = 1.0000001f, ad = seed * 1e-9f;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < nv; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m), "v"(ad));
        }
        t1 = __builtin_readcyclecounter();
        acc_out = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
    }
    if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc_out == 12345.678f) sink[threadIdx.x] = acc_out;
}

// the SAME wave: every matrix instruction followed by NF independent v_fma_f32 (one wave per SIMD: waves 4-7 exit)
template <int NF>
__global__ void __launch_bounds__(512, 2) probe_same(int nm, float seed, unsigned long long* cycles, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) return;
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = seed + k + lane;
    const float a = seed + lane, b = seed * 0.5f + lane, m = 1.0000001f, ad = seed * 1e-9f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < nm; i += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[k]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(k * NF + f) & 7]) : "v"(m), "v"(ad));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    const float acc_out = c[0][0] + c[1][1] + c[2][2] + c[3][3] + v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
    if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc_out == 12345.678f) sink[threadIdx.x] = acc_out;
}

template <int NF> void run_same(int nm) {
    unsigned long long* d_c; float* d_s;
    const int nb = 256;
    (void)hipMalloc(&d_c, nb * 8 * 8); (void)hipMalloc(&d_s, 4096);
    std::vector<unsigned long long> h(nb * 8);
    for (int rep = 0; rep < 3; ++rep) probe_same<NF><<<nb, 512>>>(nm, 1.0f, d_c, d_s);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_c, nb * 8 * 8, hipMemcpyDeviceToHost);
    std::vector<double> m;
    for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) m.push_back((double)h[b * 8 + w]);
    std::sort(m.begin(), m.end());
    printf("same wave: f32 16x16x4 + %d v_fma_f32 each: %.1f cycles per matrix instruction\n", NF, m[m.size() / 2] / nm);
    (void)hipFree(d_c); (void)hipFree(d_s);
}

// cross-wave with priorities and a matrix wave that leaves gaps: waves 0-3 issue one matrix instruction, then s_nop padding
template <int PAD, int PRIO_V>
__global__ void __launch_bounds__(512, 2) probe_gap(int nm, int nv, int run_m, int run_v, float seed, unsigned long long* cycles, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0 = 0, t1 = 0;
    float acc_out = 0.f;
    __syncthreads();
    if (wave < 4) {
        if (run_m) {
            f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const float a = seed + lane, b = seed * 0.5f + lane;
            t0 = __builtin_readcyclecounter();
            for (int i = 0; i < nm; i += 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[k]) : "v"(a), "v"(b));
#pragma unroll
                    for (int q = 0; q < PAD; ++q) asm volatile("s_nop 7");
                }
            }
            acc_out = c[0][0] + c[1][1] + c[2][2] + c[3][3];
            t1 = __builtin_readcyclecounter();
        }
    } else if (run_v) {
        __builtin_amdgcn_s_setprio(PRIO_V);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = seed + k + lane;
        const float m = 1.0000001f, ad = seed * 1e-9f;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < nv; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m), "v"(ad));
        }
        t1 = __builtin_readcyclecounter();
        acc_out = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
    }
    if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc_out == 12345.678f) sink[threadIdx.x] = acc_out;
}

template <int PAD, int PRIO_V> void run_gap(int nm, int nv) {
    unsigned long long* d_c; float* d_s;
    const int nb = 256;
    (void)hipMalloc(&d_c, nb * 8 * 8); (void)hipMalloc(&d_s, 4096);
    std::vector<unsigned long long> h(nb * 8);
    auto go = [&](int rm, int rv, double& cm, double& cv) {
        (void)hipMemset(d_c, 0, nb * 8 * 8);
        for (int rep = 0; rep < 3; ++rep) probe_gap<PAD, PRIO_V><<<nb, 512>>>(nm, nv, rm, rv, 1.0f, d_c, d_s);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d_c, nb * 8 * 8, hipMemcpyDeviceToHost);
        std::vector<double> m, v;
        for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v).push_back((double)h[b * 8 + w]);
        std::sort(m.begin(), m.end()); std::sort(v.begin(), v.end());
        cm = m[m.size() / 2]; cv = v[v.size() / 2];
    };
    double m_alone, v_alone, m_both, v_both, d;
    go(1, 0, m_alone, d); go(0, 1, d, v_alone); go(1, 1, m_both, v_both);
    printf("padding %d x s_nop 7, vector wave at priority %d: matrix wave %.0f alone -> %.0f together (%.1f per instruction); vector wave %.0f alone -> %.0f together\n",
           PAD, PRIO_V, m_alone, m_both, m_both / nm, v_alone, v_both);
    (void)hipFree(d_c); (void)hipFree(d_s);
}

template <int KIND> void run(const char* name, int nm, int nv) {
    unsigned long long* d_c; float* d_s;
    const int nb = 256;
    hipMalloc(&d_c, nb * 8 * 8); hipMalloc(&d_s, 4096);
    std::vector<unsigned long long> h(nb * 8);
    auto go = [&](int rm, int rv, double& cm, double& cv) {
        hipMemset(d_c, 0, nb * 8 * 8);
        for (int rep = 0; rep < 3; ++rep) probe<KIND><<<nb, 512>>>(nm, nv, rm, rv, 1.0f, d_c, d_s);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_c, nb * 8 * 8, hipMemcpyDeviceToHost);
        std::vector<double> m, v;
        for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v).push_back((double)h[b * 8 + w]);
        std::sort(m.begin(), m.end()); std::sort(v.begin(), v.end());
        cm = m[m.size() / 2]; cv = v[v.size() / 2];
    };
    double m_alone, v_alone, m_both, v_both, d;
    go(1, 0, m_alone, d); go(0, 1, d, v_alone); go(1, 1, m_both, v_both);
    printf("%s: %d matrix instructions: %.0f cycles alone (%.1f each), %.0f beside the vector wave (%.1f each)\n", name, nm, m_alone,
           m_alone / nm, m_both, m_both / nm);
    printf("%s: %d v_fma_f32:          %.0f cycles alone (%.2f each), %.0f beside the matrix wave (%.2f each)\n", name, nv, v_alone,
           v_alone / nv, v_both, v_both / nv);
    hipFree(d_c); hipFree(d_s);
}

int main() {
    // equal alone-durations, so that "both" shows overlap (same time) or turn-taking (the sum)
    run<0>("f32 16x16x4 ", 4096, 4096 * 8);      // 4096 x 32 cycles = 131 k; 32768 x 4 cycles = 131 k
    run<0>("f32 16x16x4 ", 4096, 4096 * 4);      // the vector wave half as long
    run<1>("bf16 16x16x16", 16384, 4096 * 8);    // 16384 x 8 = 131 k
    run_same<0>(4096); run_same<2>(4096); run_same<4>(4096); run_same<5>(4096); run_same<6>(4096); run_same<8>(4096);
    run_gap<0, 0>(4096, 16384); run_gap<0, 1>(4096, 16384); run_gap<0, 3>(4096, 16384);
    run_gap<4, 0>(4096, 16384); run_gap<4, 3>(4096, 16384);
    return 0;
}
