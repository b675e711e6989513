// zp.hpp -- GF(p) arithmetic on balanced representatives, host + device (gfx950).
//
// Contract (reference src/SpaSM.jl:73-88, :383-390): elements are i32 in [mhalfp, halfp],
// mhalfp = p/2 - p + 1, halfp = p/2, 2 < p <= 0xfffffffb.  Every function that returns a field
// element returns THE canonical balanced residue, so any correct reduction yields the same bits as
// libspasm's float-quotient formula (q = round(a*b/p); a*b - q*p; normalize).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define ZP_HD __host__ __device__ __forceinline__
#define ZP_D __device__ __forceinline__
#else
#define ZP_HD inline
#endif

struct ZpField {
    int64_t p;
    int64_t halfp;
    int64_t mhalfp;
    double dinvp;
    float finvp;
    int small;   // 1 when p < 2^16: products of two residues fit in i32 and lazy i32 accumulation is used
};

inline ZpField zp_field_make(int64_t p)
{
    ZpField F;
    F.p = p;
    F.halfp = p / 2;
    F.mhalfp = p / 2 - p + 1;
    F.dinvp = 1.0 / (double)p;
    F.finvp = (float)(1.0 / (double)p);
    F.small = p < 65536 ? 1 : 0;
    return F;
}

ZP_HD int zp_normalize(const ZpField &F, int64_t x)
{
    if (x < F.mhalfp) x += F.p;
    else if (x > F.halfp) x -= F.p;
    return (int)x;
}

// full reduction of an arbitrary i64 (|x| < 2^62) to the balanced residue
ZP_HD int zp_reduce(const ZpField &F, int64_t x)
{
    int64_t q = (int64_t)rint((double)x * F.dinvp);
    int64_t r = x - q * F.p;
    // q is within 1 of the exact quotient, so one correction either way is enough
    if (r < F.mhalfp) r += F.p;
    else if (r > F.halfp) r -= F.p;
    if (r < F.mhalfp) r += F.p;
    else if (r > F.halfp) r -= F.p;
    return (int)r;
}

ZP_HD int zp_add(const ZpField &F, int a, int b) { return zp_normalize(F, (int64_t)a + (int64_t)b); }
ZP_HD int zp_sub(const ZpField &F, int a, int b) { return zp_normalize(F, (int64_t)a - (int64_t)b); }
ZP_HD int zp_neg(const ZpField &F, int a) { return zp_normalize(F, -(int64_t)a); }

ZP_HD int zp_mul(const ZpField &F, int a, int b)
{
    int64_t q = (int64_t)rint((double)a * (double)b * F.dinvp);
    return zp_normalize(F, (int64_t)a * (int64_t)b - q * F.p);
}

// a*x + y
ZP_HD int zp_axpy(const ZpField &F, int a, int x, int y)
{
    int64_t q = (int64_t)rint(((double)a * (double)x + (double)y) * F.dinvp);
    return zp_normalize(F, (int64_t)a * (int64_t)x + (int64_t)y - q * F.p);
}

// modular inverse by the extended Euclidean algorithm (gcdx in src/SpaSM.jl:386)
ZP_HD int zp_inverse(const ZpField &F, int a)
{
    int64_t r0 = a < 0 ? (int64_t)a + F.p : (int64_t)a, r1 = F.p;
    int64_t s0 = 1, s1 = 0;
    while (r1 != 0) {
        int64_t q = r0 / r1;
        int64_t t = r0 - q * r1; r0 = r1; r1 = t;
        t = s0 - q * s1; s0 = s1; s1 = t;
    }
    int64_t r = s0 % F.p;
    if (r < 0) r += F.p;
    return zp_normalize(F, r);
}

#if defined(__HIPCC__)
// Lazy product for accumulation: returns a value congruent to a*b mod p, NOT canonical.
//   SMALL (p < 2^16): |a*b| < 2^30 fits i32; one float-quotient step leaves |r| <= 0.51 p < 2^15.1,
//                     so an i32 accumulator absorbs > 60000 terms before the final zp_reduce.
//   general:          i64 product, double quotient, |r| <= 0.51 p < 2^31.1; accumulate in i64.
template <bool SMALL> struct ZpAcc;
// x - rn(x/p) * p for p < 2^16 and |x/p| < 2^22 in four full-rate instructions: the quotient is read off the mantissa of
// x * (-1/p) + 1.5 * 2^23 (one rounding, to the integer grid), whose low 24 bits are 2^22 - q; the bias 2^22 * p is folded
// into the addend (arithmetic mod 2^32).  |result| <= p/2 + |x/p| * p * 2^-22.
ZP_D int zp_small_lazy(int x, float ninvp, int p)
{
    const float g = __fmaf_rn((float)x, ninvp, 12582912.0f);
    return __mul24(__float_as_int(g), p) + (x - (int)((unsigned)p << 22)); // v_mad_i32_i24
}

template <> struct ZpAcc<true> {
    typedef int type;
    static ZP_D int mul_lazy(const ZpField &F, int a, int b)
    {
        // balanced residues of a prime < 2^16 are 24-bit operands: full-rate v_mul_i32_i24, |quotient| <= p/4 + 1
        return zp_small_lazy(__mul24(a, b), -F.finvp, (int)F.p);
    }
};
template <> struct ZpAcc<false> {
    typedef long long type;
    static ZP_D long long mul_lazy(const ZpField &F, int a, int b)
    {
        long long prod = (long long)a * (long long)b;
        long long q = (long long)rint((double)a * (double)b * F.dinvp);
        return prod - q * F.p;
    }
};
#endif
