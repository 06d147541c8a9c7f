// The counter-based dropout generator of the head kernels (kgnn_readout.hip) and of the fused tail (kgnn_tail.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mkgnn {

// Philox4x32-10 (Salmon et al., SC'11): counter-based, so the backward regenerates the forward's mask instead of
// storing it.  counter = (element / 4, offset), key = seed; element e takes word e % 4.
__device__ __forceinline__ uint32_t philox_word(uint64_t seed, uint64_t offset, uint64_t element) {
    uint32_t c0 = (uint32_t)(element >> 2), c1 = (uint32_t)(element >> 34), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t w[4] = {c0, c1, c2, c3};
    return w[element & 3];
}
// dropout multiplier of element e: 0 with probability p, else 1 / (1 - p)
__device__ __forceinline__ float keep_scale_of(uint64_t seed, uint64_t offset, uint64_t element, float p) {
    const float u = (float)(philox_word(seed, offset, element) >> 8) * (1.f / 16777216.f);     // [0, 1)
    return u >= p ? 1.f / (1.f - p) : 0.f;
}

}  // namespace mkgnn
