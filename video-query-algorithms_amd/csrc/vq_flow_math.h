// Correctly rounded fp32 division and square root on TWO values per lane, written out so that their refinement steps run as packed
// instructions (csrc/vq_flow.hip: the TV-L1 iteration).  Bit for bit the results of `a / b` and sqrtf() as this compiler expands them
// (checked on 2^30 operand pairs incl. denormals, zeros, infinities and NaNs: tools/ubench/flow_math_check.hip, tests/test_flow_gpu.py).
#pragma once
#include <hip/hip_runtime.h>

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 div_ieee(f2 n, f2 d) {
    bool s0, s1, t0, t1;
    const f2 ds = {__builtin_amdgcn_div_scalef(n.x, d.x, false, &s0), __builtin_amdgcn_div_scalef(n.y, d.y, false, &s1)};
    const f2 ns = {__builtin_amdgcn_div_scalef(n.x, d.x, true, &t0), __builtin_amdgcn_div_scalef(n.y, d.y, true, &t1)};
    const f2 rc = {__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y)};
    const f2 one = {1.0f, 1.0f};
    const f2 e0 = __builtin_elementwise_fma(-ds, rc, one);
    const f2 r1 = __builtin_elementwise_fma(e0, rc, rc);
    const f2 q0 = ns * r1;
    const f2 e1 = __builtin_elementwise_fma(-ds, q0, ns);
    const f2 q1 = __builtin_elementwise_fma(e1, r1, q0);
    const f2 e2 = __builtin_elementwise_fma(-ds, q1, ns);
    return f2{__builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e2.x, r1.x, q1.x, t0), d.x, n.x),
              __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e2.y, r1.y, q1.y, t1), d.y, n.y)};
}

// sqrtf(x) for x >= +0 (a sum of squares): v_sqrt_f32 (1 ulp) and the library's correction -- the neighbours one ulp down and up, their
// residuals x - n * s by FMA, two selects (AMDGPUTargetLowering::lowerFSQRTF32).  The library scales x < 2^-96 by 2^32 first and returns x
// itself for zeros and infinities; for x = +0 and x >= 2^-96 those steps select what the plain sequence computes, so only a lane with
// 0 < x < 2^-96 (a gradient below 3.6e-15 that is not zero) sends its wave through the library's full sequence.
__device__ __forceinline__ f2 sqrt_ieee(f2 x) {
    const unsigned bx = __float_as_uint(x.x) - 1u, by = __float_as_uint(x.y) - 1u;       // +0 -> 0xFFFFFFFF: not below the bound
    if (__builtin_expect(bx < 0x0F7FFFFFu || by < 0x0F7FFFFFu, 0)) return f2{sqrtf(x.x), sqrtf(x.y)};
    const f2 s = {__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
    const f2 dn = {__uint_as_float(__float_as_uint(s.x) - 1u), __uint_as_float(__float_as_uint(s.y) - 1u)};
    const f2 up = {__uint_as_float(__float_as_uint(s.x) + 1u), __uint_as_float(__float_as_uint(s.y) + 1u)};
    const f2 rd = __builtin_elementwise_fma(-dn, s, x), ru = __builtin_elementwise_fma(-up, s, x);
    f2 r;
    r.x = rd.x <= 0.0f ? dn.x : s.x;
    r.y = rd.y <= 0.0f ? dn.y : s.y;
    r.x = ru.x > 0.0f ? up.x : r.x;
    r.y = ru.y > 0.0f ? up.y : r.y;
    return r;
}
