// LDS-DMA gather throughput by request shape (diagnostics): how fast can a CU pull 16 x 448-byte rows (random row ids) into
// LDS when a 1 KB piece is (A) 16 rows x one 64-byte segment each (the forward's MFMA-operand layout), (B) 64 consecutive
// 16-byte chunks of the rows (row-major: the bank kernel's layout), (C) 64 consecutive 16-byte chunks of ONE contiguous 1 KB
// block (an upper bound)?  hipcc --offload-arch=gfx950 -O3 tools/probes/dma_probe.hip -o gpurun_out/dma_probe && ./dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ void dma16(const void* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int MODE, int INFLIGHT>
__global__ void __launch_bounds__(256, 2) probe(const float* __restrict__ x, const int* __restrict__ ids, int n_ids, int iters,
                                                unsigned long long* out) {
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* buf = lds + wave * (INFLIGHT * 256);
    const int gw = blockIdx.x * 4 + wave;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int tile = (gw * iters + it) % (n_ids / 16);
#pragma unroll
        for (int p = 0; p < INFLIGHT; ++p) {
            const int piece = p % 7;
            const float* src;
            if (MODE == 0) {            // 16 rows x 64 B: lane (row = lane & 15, quarter = lane >> 4) -> chunk 4 piece + quarter
                const int row = ids[tile * 16 + (lane & 15)];
                src = x + (size_t)row * 112 + 16 * piece + 4 * (lane >> 4);
            } else if (MODE == 1) {     // row-major: chunk g = 64 piece + lane of the 16 x 28 chunk image
                const int g = 64 * piece + lane, r = g / 28, c = g - r * 28;
                const int row = ids[tile * 16 + r];
                src = x + (size_t)row * 112 + 4 * c;
            } else {                    // one contiguous KB
                const int row = ids[tile * 16 + (p & 15)];
                src = x + (size_t)row * 112 + 4 * (lane % 28) + (lane / 28) * 112;
            }
            dma16(src, buf + p * 256);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[gw] = t1 - t0;
    if (lds[threadIdx.x] == 123.456f) out[0] = 0;   // keep the LDS alive
}

int main() {
    const int n_rows = 400000;            // 179 MB of rows
    std::vector<int> ids(1 << 20);
    srand(1);
    // ids like a molecule batch's neighbour lists: near-sequential with jitter
    for (size_t i = 0; i < ids.size(); ++i) ids[i] = (int)((i * 37 / 100 + rand() % 64) % n_rows);
    float* x; int* d_ids; unsigned long long* out;
    hipMalloc(&x, (size_t)n_rows * 112 * 4); hipMemset(x, 0, (size_t)n_rows * 112 * 4);
    hipMalloc(&d_ids, ids.size() * 4); hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 2048 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    auto run = [&](const char* name, auto kernel, int inflight) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            kernel<<<512, 256, 4 * inflight * 1024, 0>>>(x, d_ids, (int)ids.size(), iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) {
                const double bytes = 2048.0 * iters * inflight * 1024.0;
                printf("%-34s in flight %2d pieces/wave: %.1f us, %.2f TB/s, %.0f cycles per piece per CU-slot\n", name, inflight, ms * 1e3,
                       bytes / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (iters * inflight));
            }
        }
    };
    run("A 16 rows x 64 B segments", probe<0, 7>, 7);
    run("B row-major 64 chunks", probe<1, 7>, 7);
    run("C contiguous KB", probe<2, 7>, 7);
    run("A 16 rows x 64 B segments", probe<0, 14>, 14);
    run("B row-major 64 chunks", probe<1, 14>, 14);
    run("C contiguous KB", probe<2, 14>, 14);
    run("A 16 rows x 64 B segments", probe<0, 28>, 28);
    run("B row-major 64 chunks", probe<1, 28>, 28);
    return 0;
}
