// demcz_device.h -- device-side arithmetic of the DEMCz chain update for gfx950.
//
// Implements the arithmetic spec of DESIGN.md section 3: IEEE binary64, round-to-nearest-even,
// no contraction (this file is compiled with -ffp-contract=off; fused multiply-adds appear only
// where fma() is written).  +,-,*,/ and sqrt are correctly rounded on gfx950, so a host
// implementation of the same operation sequence agrees bit for bit.
//
// RNG: rocRAND's Philox4x32-10 device engine.  Global chain c owns subsequence c; block-step
// draws are taken sequentially from it, a whole number of 4-word blocks each, so the stream
// position of generation g is a closed form and a run can be cut into windows anywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace demcz {

typedef rocrand_state_philox4x32_10 rng_state;

// Position the stream of `chain` at 4-word block `blk`.
__device__ __forceinline__ void rng_seek(rng_state& st, uint64_t seed, uint64_t chain, uint64_t blk)
{
    rocrand_init(seed, chain, 4ull * blk, &st);
}

// Next 4-word block as two 64-bit words (little-endian pairs).
__device__ __forceinline__ void rng_next(rng_state& st, uint64_t& r1, uint64_t& r2)
{
    uint4 w = rocrand4(&st);
    r1 = (uint64_t)w.x | ((uint64_t)w.y << 32);
    r2 = (uint64_t)w.z | ((uint64_t)w.w << 32);
}

// ((r >> 12) + 0.5) * 2^-52 in (0,1), exact.
__device__ __forceinline__ double u_open(uint64_t r)
{
    return ((double)(r >> 12) + 0.5) * 0x1p-52;
}

// Natural log of a positive normal double: x = 2^k (1+f), sqrt(1/2) < 1+f <= sqrt(2),
// log(1+f) = f - f^2/2 + s (f^2/2 + R(s^2)), s = f/(2+f).  < 1 ulp.
__device__ __forceinline__ double dm_log(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double LG1 = 6.666666666666735130e-01;
    const double LG2 = 3.999999999940941908e-01;
    const double LG3 = 2.857142874366239149e-01;
    const double LG4 = 2.222219843214978396e-01;
    const double LG5 = 1.818357216161805012e-01;
    const double LG6 = 1.531383769920937332e-01;
    const double LG7 = 1.479819860511658591e-01;
    uint64_t b = (uint64_t)__double_as_longlong(x);
    uint32_t hx = (uint32_t)(b >> 32);
    uint32_t lx = (uint32_t)b;
    hx += 0x3ff00000u - 0x3fe6a09eu;
    int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    double xr = __longlong_as_double((long long)(((uint64_t)hx << 32) | lx));
    double f = xr - 1.0;
    double hfsq = (0.5 * f) * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (LG2 + w * (LG4 + w * LG6));
    double t2 = z * (LG1 + w * (LG3 + w * (LG5 + w * LG7)));
    double R = t2 + t1;
    double dk = (double)k;
    return ((((s * (hfsq + R)) + (dk * LN2_LO)) - hfsq) + f) + (dk * LN2_HI);
}

// (cos, sin)(2 pi k53 / 2^53): quadrant by integer arithmetic, |theta| <= pi/4 kernels.
__device__ __forceinline__ void dm_sincos2pi(uint64_t k53, double& c, double& s)
{
    const double TWO_PI_HI = 6.28318530717958623200e+00;
    const double TWO_PI_LO = 2.44929359829470641435e-16;
    const double S1 = -1.66666666666666324348e-01;
    const double S2 = 8.33333333332248946124e-03;
    const double S3 = -1.98412698298579493134e-04;
    const double S4 = 2.75573137070700676789e-06;
    const double S5 = -2.50507602534068634195e-08;
    const double S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02;
    const double C2 = -1.38888888888741095749e-03;
    const double C3 = 2.48015872894767294178e-05;
    const double C4 = -2.75573143513906633035e-07;
    const double C5 = 2.08757232129817482790e-09;
    const double C6 = -1.13596475577881948265e-11;
    uint64_t q = (k53 + (1ull << 50)) >> 51;
    long long kt = (long long)k53 - (long long)(q << 51);
    double t = (double)kt * 0x1p-53;
    double th = fma(t, TWO_PI_LO, t * TWO_PI_HI);
    double z = th * th;
    double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double v = z * th;
    double sn = th + v * (S1 + z * rs);
    double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double hz = 0.5 * z;
    double wv = 1.0 - hz;
    double cs = wv + (((1.0 - wv) - hz) + (z * rc));
    unsigned qq = (unsigned)q & 3u;
    // q: 0 -> (cs, sn); 1 -> (-sn, cs); 2 -> (-cs, -sn); 3 -> (sn, -cs)
    double a = (qq & 1u) ? sn : cs;
    double bb = (qq & 1u) ? cs : sn;
    c = (qq == 1u || qq == 2u) ? -a : a;
    s = (qq >= 2u) ? -bb : bb;
}

// Box-Muller pair from one 4-word block.
__device__ __forceinline__ void normal_pair(uint64_t r1, uint64_t r2, double& z0, double& z1)
{
    double lg = dm_log(u_open(r1));
    double R = sqrt(-2.0 * lg);
    double c, s;
    dm_sincos2pi(r2 >> 11, c, s);
    z0 = R * c;
    z1 = R * s;
}

// floor(r * m / 2^64) for m < 2^32: r = h 2^32 + l  =>  floor((h m + floor(l m / 2^32)) / 2^32), exact.
// Two multiplies instead of the 64x64->128 product (the archive has fewer than 2^32 rows in every
// configuration of interest; the general form stays for larger ones).
__device__ __forceinline__ uint64_t mulhi64_u32(uint64_t r, uint32_t m)
{
    const uint64_t t = (uint64_t)(uint32_t)(r >> 32) * m + __umulhi((uint32_t)r, m);
    return t >> 32;
}

// Two distinct archive rows out of M: i1 ~ U{0..M-1}, i2 ~ U({0..M-1} \ {i1}).
__device__ __forceinline__ void draw_rows(uint64_t r1, uint64_t r2, uint64_t M, uint64_t& i1, uint64_t& i2)
{
    uint64_t j;
    if (M <= 0xffffffffull) {           // wave-uniform
        i1 = mulhi64_u32(r1, (uint32_t)M);
        j = mulhi64_u32(r2, (uint32_t)(M - 1));
    } else {
        i1 = __umul64hi(r1, M);
        j = __umul64hi(r2, M - 1);
    }
    i2 = j + (j >= i1 ? 1ull : 0ull);
}

}  // namespace demcz
