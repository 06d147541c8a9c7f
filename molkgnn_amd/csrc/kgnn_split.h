// Exact fp16 splits of fp32 operands for the matrix pipe (gfx950) -- shared by the streamed forward (kgnn_fwd_stream.hip,
// BF = 2) and the streamed backward kernels (kgnn_bwd_rows_stream.hip, kgnn_bwd_stream.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mkgnn {

typedef float split_f32x4 __attribute__((ext_vector_type(4)));
// (bit casts that exist on the host too: the power-of-two scale functions below are plain integer arithmetic on exponent fields,
// and tests/test_host_cpu.py checks them exhaustively through mkgnn_debug_split_scales without a GPU)
__host__ __device__ __forceinline__ float split_bits_to_float(uint32_t u) { return __builtin_bit_cast(float, u); }
__host__ __device__ __forceinline__ uint32_t split_float_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// fp32 products out of fp16 matrix instructions (round 5).  A float is split into hi = fp16(x), lo = fp16(x - hi): two roundings
// to nearest with unit roundoff 2^-11 (fp16 keeps 11 significant bits).  x - hi is exact in fp32 and at most 2^-11 |x|; it has up
// to 13 significant bits of which lo keeps 11, so
//     |x - hi - lo| <= 2^-11 |x - hi| <= 2^-22 |x|                    (lo a normal fp16 number; hi + lo is exact in fp32)
// (round 5's header said 2^-24: wrong by a factor of four -- VERDICT round 5; the measured error never depended on it).  A
// product x y is then hi hi' + hi lo' + lo hi': the dropped lo lo' is at most 2^-22 |x y|, each operand's own residual adds
// 2^-22 |x y|, so one product is off by at most 3 * 2^-22 |x y|, and a dot product of two rows by at most
// 3 * 2^-22 sum |x_i y_i| <= 3 * 2^-22 |x| |y| (7e-7 for unit rows; a worst case -- the residuals are rounding errors of either
// sign and the measured figure is 1.2e-7, below the fp32 fma chain's 2.3e-7: tests/test_scale_parity.py) plus the fp32
// accumulation both forms share.  Three v_mfma_f32_16x16x16_f16 of 8 cycles in place of four v_mfma_f32_16x16x4_f32 of 32.
// All of it provided nothing under- or overflows in fp16 (normal numbers: 2^-14 .. 2^16): every user scales its operands by
// exact powers of two first and takes them out of the result again -- the forward per atom row and per bank row
// (kgnn_fwd_stream.hip, BF = 2), the rows kernel per atom (kgnn_bwd_rows_stream.hip), the bank kernel per atom row and per wave
// (kgnn_bwd_stream.hip).  An ELEMENT far below its row's scale loses its lo half to fp16 subnormals (spacing 2^-24): below 2^-3
// after scaling -- 2^-11 of the row's norm -- its error is 2^-25 absolute, i.e. 2^-33 of the row's norm: invisible in a cosine.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
struct SplitReg { h16x4 hi, lo; };
__device__ __forceinline__ SplitReg split_f16(split_f32x4 v) {
    SplitReg r;
    r.hi = h16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    r.lo = h16x4{(_Float16)(v[0] - (float)r.hi[0]), (_Float16)(v[1] - (float)r.hi[1]), (_Float16)(v[2] - (float)r.hi[2]), (_Float16)(v[3] - (float)r.hi[3])};
    return r;
}
// The split of v * s (s a power of two) in ten instructions per four values: two v_pk_mul_f32, two v_cvt_pk_f16_f32 (hi), four
// v_fma_mix_f32 (r = v s - hi with hi taken straight from its fp16 half: exact), two v_cvt_pk_f16_f32 (lo).  Left to the compiler
// the residual alone is a v_cvt_f32_f16 and a subtraction per value (17 instructions), and on this chip vector instructions do
// not hide behind the other wave's matrix instructions.  Only the v_fma_mix_f32 is inline asm, and what it
// writes is read by ordinary vector instructions: the operands of the matrix instructions come out of compiler-visible
// v_cvt_pk_f16_f32.  (A first version built hi and lo with v_fma_mixlo_f16 / v_fma_mixhi_f16 -- eight instructions -- and was
// WRONG on the hardware: a matrix instruction that reads a register a few cycles after a 16-bit partial write of an inline-asm
// instruction gets the old half; the hazard recognizer puts one wait state there, eight made it right.  Found by
// tests/test_scale_parity.py::test_split_fp16_products_are_fp32_grade.)
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float residual_lo(float xs, h16x2 hi) {     // xs - (float)hi[0]
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(xs), "v"(hi));
    return r;
}
__device__ __forceinline__ float residual_hi(float xs, h16x2 hi) {     // xs - (float)hi[1]
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(xs), "v"(hi));
    return r;
}
__device__ __forceinline__ SplitReg split_exact(split_f32x4 xs) {         // (the caller has scaled: eight instructions)
    const h16x2 h0 = {(_Float16)xs[0], (_Float16)xs[1]}, h1 = {(_Float16)xs[2], (_Float16)xs[3]};
    const h16x2 l0 = {(_Float16)residual_lo(xs[0], h0), (_Float16)residual_hi(xs[1], h0)};
    const h16x2 l1 = {(_Float16)residual_lo(xs[2], h1), (_Float16)residual_hi(xs[3], h1)};
    SplitReg r;
    r.hi = h16x4{h0[0], h0[1], h1[0], h1[1]};
    r.lo = h16x4{l0[0], l0[1], l1[0], l1[1]};
    return r;
}
__device__ __forceinline__ SplitReg split_scaled(split_f32x4 v, float s) { return split_exact(v * s); }
__device__ __forceinline__ SplitReg split_scaled(split_f32x4 v, split_f32x4 s) { return split_exact(v * s); }      // a scale per value

// ---- pre-split rows (round 6) ------------------------------------------------------------------------------------------
// A row that only the streamed kernels read -- h = propagate(sim_sc) between two kernel convolutions, the batch norm's output in
// front of the first -- is written by its producer as the operand the matrix instructions take: four consecutive floats x[4 g ..
// 4 g + 3] become the sixteen bytes  hi(0..3) | lo(0..3)  (fp16 each) of x * s, s = 2^(exponent(1 / max(|row|, eps)) + 8): the scale
// the forward and the bank kernel applied per wave and per role until round 5 (ten vector instructions per sixteen floats, in
// every wave that touched the row).  Same bytes per row, same 16-byte chunks: DMA pieces, LDS images and swizzles are untouched.
// The forward's products are bit for bit what they were (the same split of the same scaled value).  A reader that wants the
// value back takes (hi + lo) / s: exact in fp32 (hi holds the top 11 bits, lo the next 11 of the 13 that remain), i.e. x to
// 2^-22 |x| -- the gather that undoes the row normalisation and the raw-row equality test of the chirality branch.
constexpr int SPLIT_ROW_EXP_BITS = 8;
__host__ __device__ __forceinline__ float split_row_scale_of(float inv) {  // 2^(exponent(inv) + 8)   (inv <= 1e8: no overflow)
    return split_bits_to_float((split_float_to_bits(inv) & 0x7f800000u) + ((uint32_t)SPLIT_ROW_EXP_BITS << 23));
}
// inv / scale: inv's mantissa with the exponent -8
__host__ __device__ __forceinline__ float split_row_inv(float inv) {
    return split_bits_to_float((split_float_to_bits(inv) & 0x007fffffu) | ((uint32_t)(127 - SPLIT_ROW_EXP_BITS) << 23));
}
// four floats of a row -> their sixteen bytes in the pre-split form (returned as the four dwords to store)
__device__ __forceinline__ split_f32x4 split_row_store(split_f32x4 v, float inv) {
    const SplitReg r = split_exact(v * split_row_scale_of(inv));
    return __builtin_bit_cast(split_f32x4, r);
}
// ... and back: hi + lo (the SCALED values; multiply by split_row_inv(inv) for x / |x|, divide by the scale for x)
__device__ __forceinline__ split_f32x4 split_row_value(split_f32x4 w) {
    const SplitReg r = __builtin_bit_cast(SplitReg, w);
    return split_f32x4{(float)r.hi[0] + (float)r.lo[0], (float)r.hi[1] + (float)r.lo[1], (float)r.hi[2] + (float)r.lo[2],
                       (float)r.hi[3] + (float)r.lo[3]};
}

// three matrix instructions for one split product tile: acc += a b with a, b split (lo lo' dropped: at most 2^-22 of |a b|)
__device__ __forceinline__ split_f32x4 split_mfma(const SplitReg& a, const SplitReg& b, split_f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.hi, acc, 0, 0, 0);
    return acc;
}

// Power-of-two scale that brings a magnitude `amax` to [2^TARGET, 2^(TARGET+1)) and its reciprocal times 2^-EXTRA, both exact
// (exponent-field arithmetic; amax = 0 or tiny: the scale saturates at 2^103, still a power of two, still undone exactly)
template <int TARGET> __host__ __device__ __forceinline__ float split_scale_for(float amax) {
    const int eb = (int)((split_float_to_bits(amax) >> 23) & 0xffu);      // biased exponent of amax
    int f = 254 + TARGET - eb;                                             // biased exponent of 2^(TARGET - e)
    f = f > 230 ? 230 : (f < 24 ? 24 : f);
    return split_bits_to_float((uint32_t)f << 23);
}
// ... from an upper bound of the magnitude's biased exponent
template <int TARGET> __host__ __device__ __forceinline__ float split_scale_for_exponent(int eb) {
    int f = 254 + TARGET - eb;
    f = f > 230 ? 230 : (f < 24 ? 24 : f);
    return split_bits_to_float((uint32_t)f << 23);
}
template <int EXTRA> __host__ __device__ __forceinline__ float split_unscale_of(float scale) {      // 2^-EXTRA / scale
    const int f = (int)(split_float_to_bits(scale) >> 23);
    return split_bits_to_float((uint32_t)(254 - EXTRA - f) << 23);
}

}  // namespace mkgnn
