// Probe: where does global_load_lds_dwordx4 / _dword with an instruction offset land in LDS, and what does it read?
// (kgnn_fwd_stream.hip gives every LDS-DMA site its own immediate offset so that the compiler cannot merge two sites
// into one instruction; that is only right if the offset is added on BOTH sides: memory address and LDS address.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int OFF, int SIZE>
__global__ void probe(const float* src, float* out) {
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = -1.f;
    __syncthreads();
    // intended: LDS floats [256 + lane * SIZE/4 ..) <- src[1000 + lane * SIZE/4 ..)
    const char* s = (const char*)(src + 1000 + lane * (SIZE / 4)) - OFF;
    char* d = (char*)(lds + 256) - OFF;
#define GP (const __attribute__((address_space(1))) void*)s
#define LP (__attribute__((address_space(3))) void*)d
    if constexpr (SIZE == 16 && OFF == 0) __builtin_amdgcn_global_load_lds(GP, LP, 16, 0, 0);
    else if constexpr (SIZE == 16 && OFF == 16) __builtin_amdgcn_global_load_lds(GP, LP, 16, 16, 0);
    else if constexpr (SIZE == 16 && OFF == 32) __builtin_amdgcn_global_load_lds(GP, LP, 16, 32, 0);
    else if constexpr (SIZE == 16 && OFF == 48) __builtin_amdgcn_global_load_lds(GP, LP, 16, 48, 0);
    else if constexpr (SIZE == 4 && OFF == 0) __builtin_amdgcn_global_load_lds(GP, LP, 4, 0, 0);
    else if constexpr (SIZE == 4 && OFF == 4) __builtin_amdgcn_global_load_lds(GP, LP, 4, 4, 0);
    else if constexpr (SIZE == 4 && OFF == 8) __builtin_amdgcn_global_load_lds(GP, LP, 4, 8, 0);
    else if constexpr (SIZE == 4 && OFF == 20) __builtin_amdgcn_global_load_lds(GP, LP, 4, 20, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}

template <int OFF, int SIZE> int run(const float* d_src, float* d_out) {
    probe<OFF, SIZE><<<1, 64, 4096>>>(d_src, d_out);
    std::vector<float> h(1024);
    hipMemcpy(h.data(), d_out, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    const int n = 64 * (SIZE / 4);
    for (int i = 0; i < 1024; ++i) {
        const float want = (i >= 256 && i < 256 + n) ? (float)(1000 + (i - 256)) : -1.f;
        if (h[i] != want) { if (bad < 4) printf("  OFF %d SIZE %d: lds[%d] = %g, want %g\n", OFF, SIZE, i, h[i], want); ++bad; }
    }
    printf("OFF %3d SIZE %2d: %s (%d wrong)\n", OFF, SIZE, bad ? "DIFFERENT" : "as intended", bad);
    return bad;
}

int main() {
    float *d_src, *d_out;
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    hipMalloc(&d_src, 4096 * 4); hipMalloc(&d_out, 4096);
    hipMemcpy(d_src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    int bad = 0;
    bad += run<0, 16>(d_src, d_out); bad += run<16, 16>(d_src, d_out); bad += run<32, 16>(d_src, d_out); bad += run<48, 16>(d_src, d_out);
    bad += run<0, 4>(d_src, d_out); bad += run<4, 4>(d_src, d_out); bad += run<8, 4>(d_src, d_out); bad += run<20, 4>(d_src, d_out);
    return bad ? 1 : 0;
}
