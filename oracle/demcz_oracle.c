/*
 * demcz_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the DEMCz chain-update path of
 * chrished/DEMC.jl (reference checked out at /root/reference; citations below are
 * relative to that tree).  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this file's shared object, and only as the
 * checker / the timed CPU baseline -- never as something the product path calls.
 *
 * PARITY STATUS: "parity unpinned" at the bit level against the Julia reference.
 *   - The reference holds no golden vectors, KATs or fixtures for this path
 *     (test/ *.jl assert only R-hat < 1.1 and 0.1 < accept-ratio < 0.45), Julia is
 *     not installed here, and Julia's MersenneTwister stream cannot be regenerated.
 *   - What pins this oracle instead: Random123 Philox4x32-10 known-answer vectors,
 *     rocRAND host-stream equality, libm/mpmath accuracy checks of dm_log /
 *     dm_sincos2pi, scipy closed forms for the targets, a NumPy restatement of
 *     utils.jl:2-20 for R-hat, and the reference's own statistical predicates.
 *
 * ARITHMETIC SPEC (shared by this oracle and the HIP kernels).
 *   Everything is IEEE-754 binary64 with round-to-nearest-even, NO contraction
 *   (compile with -ffp-contract=off); fused multiply-adds appear only where written
 *   as fma().  +,-,*,/ and sqrt are correctly rounded on both the host and gfx950,
 *   so the two sides agree bit for bit.
 *
 * Reference functions restated here:
 *   update_demcz_chain_block  src/demcz.jl:174-195   -> block_step()
 *   accept (2-arg)            src/demcz.jl:197-203   -> block_step(), temperature == NULL
 *   accept (3-arg, tempered)  src/demcz_anneal.jl:172-178
 *   update_blocks             src/demcz.jl:167-172   -> loop over ib in chain_generation()
 *   runchain!                 src/demcz.jl:80-93     -> chain_generation() + append
 *   generation loop           src/demcz.jl:30-33     -> oracle_demcz_run()
 *   Rhat_gelman               src/utils.jl:2-20      -> oracle_rhat_gelman()
 *   tempbaseline              src/demcz_anneal.jl:1-3 -> oracle_tempbaseline()
 *   accept-ratio / gamma adaptation  src/demcz.jl:42, src/demcz_anneal.jl:48-57
 *   flatten_chain, mean_cov_chain    src/utils.jl:22-32, 96-111
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
 * SC'11).  Counter layout chosen to coincide with rocRAND's rocrand_state_philox4x32_10:
 *   counter = { offset/4 (64 bit, lo:hi), subsequence (64 bit, lo:hi) }, key = seed (lo:hi)
 * so that stream (seed, chain, block index b) == rocrand_init(seed, chain, 4*b) + rocrand4().
 * ------------------------------------------------------------------------------------------ */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

ORACLE_API void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* One Philox block of the stream of global chain `chain`: two 64-bit words. */
static void draw_block(uint64_t seed, uint64_t chain, uint64_t blk, uint64_t* r1, uint64_t* r2)
{
    uint32_t ctr[4] = { (uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)chain, (uint32_t)(chain >> 32) };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t w[4];
    oracle_philox4x32_10(ctr, key, w);
    *r1 = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    *r2 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
}

ORACLE_API void oracle_draw_block(uint64_t seed, uint64_t chain, uint64_t blk, uint64_t out[2])
{
    draw_block(seed, chain, blk, &out[0], &out[1]);
}

/* ------------------------------------------------------------------------------------------
 * Uniform conversions.
 *   u_open(r)  = ((r >> 12) + 0.5) * 2^-52  in (0,1), exactly representable (2k+1 < 2^53)
 *   the angle word keeps 53 bits as an integer, see dm_sincos2pi().
 * Reference uses rand() in [0,1) (demcz.jl:198); the open interval differs on a null set.
 * ------------------------------------------------------------------------------------------ */
static double u_open(uint64_t r)
{
    return ((double)(r >> 12) + 0.5) * 0x1p-52;
}

static uint64_t dbl_bits(double x) { uint64_t b; memcpy(&b, &x, 8); return b; }
static double bits_dbl(uint64_t b) { double x; memcpy(&x, &b, 8); return x; }

/* Natural log for positive normal x, classic argument reduction x = 2^k (1+f),
 * sqrt(1/2) < 1+f <= sqrt(2), log(1+f) = 2s + s*R(s^2), s = f/(2+f)
 * (degree-14 minimax in s, the coefficients W. Kahan / K.C. Ng published in FreeBSD msun
 * e_log.c).  Error < 1 ulp.  Operation order below IS the spec. */
static const double LN2_HI = 6.93147180369123816490e-01;
static const double LN2_LO = 1.90821492927058770002e-10;
static const double LG1 = 6.666666666666735130e-01;
static const double LG2 = 3.999999999940941908e-01;
static const double LG3 = 2.857142874366239149e-01;
static const double LG4 = 2.222219843214978396e-01;
static const double LG5 = 1.818357216161805012e-01;
static const double LG6 = 1.531383769920937332e-01;
static const double LG7 = 1.479819860511658591e-01;

ORACLE_API double oracle_dm_log(double x)
{
    uint64_t b = dbl_bits(x);
    uint32_t hx = (uint32_t)(b >> 32);
    uint32_t lx = (uint32_t)b;
    hx += 0x3ff00000u - 0x3fe6a09eu;
    int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    double xr = bits_dbl(((uint64_t)hx << 32) | lx);
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

/* (cos, sin)(2*pi*u) for u = k53 * 2^-53, k53 < 2^53.
 * Quadrant q = round(4u) in 0..4 by integer arithmetic, t = u - q/4 in [-1/8, 1/8] exact,
 * theta = 2*pi*t with a two-term 2*pi, then the classic |theta| <= pi/4 kernels
 * (msun k_sin.c / k_cos.c minimax coefficients), then the quadrant rotation. */
static const double TWO_PI_HI = 6.28318530717958623200e+00;  /* 0x401921FB54442D18 */
static const double TWO_PI_LO = 2.44929359829470641435e-16;  /* 2*pi - TWO_PI_HI */
static const double S1 = -1.66666666666666324348e-01;
static const double S2 = 8.33333333332248946124e-03;
static const double S3 = -1.98412698298579493134e-04;
static const double S4 = 2.75573137070700676789e-06;
static const double S5 = -2.50507602534068634195e-08;
static const double S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02;
static const double C2 = -1.38888888888741095749e-03;
static const double C3 = 2.48015872894767294178e-05;
static const double C4 = -2.75573143513906633035e-07;
static const double C5 = 2.08757232129817482790e-09;
static const double C6 = -1.13596475577881948265e-11;

ORACLE_API void oracle_dm_sincos2pi(uint64_t k53, double* cos_out, double* sin_out)
{
    uint64_t q = (k53 + ((uint64_t)1 << 50)) >> 51;          /* 0..4 */
    int64_t kt = (int64_t)k53 - (int64_t)(q << 51);           /* [-2^50, 2^50] */
    double t = (double)kt * 0x1p-53;
    double th = fma(t, TWO_PI_LO, t * TWO_PI_HI);
    double z = th * th;
    /* sin kernel */
    double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double v = z * th;
    double sn = th + v * (S1 + z * rs);
    /* cos kernel */
    double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double hz = 0.5 * z;
    double wv = 1.0 - hz;
    double cs = wv + (((1.0 - wv) - hz) + (z * rc));
    double c, s;
    switch ((int)(q & 3)) {
    case 0: c = cs; s = sn; break;
    case 1: c = -sn; s = cs; break;
    case 2: c = -cs; s = -sn; break;
    default: c = sn; s = -cs; break;
    }
    *cos_out = c;
    *sin_out = s;
}

/* Box-Muller pair from one Philox block: z0 = R cos(2 pi u2), z1 = R sin(2 pi u2),
 * R = sqrt(-2 log u1), u1 = u_open(r1), u2 = (r2 >> 11) 2^-53. */
ORACLE_API void oracle_normal_pair(uint64_t r1, uint64_t r2, double z[2])
{
    double lg = oracle_dm_log(u_open(r1));
    double R = sqrt(-2.0 * lg);
    double c, s;
    oracle_dm_sincos2pi(r2 >> 11, &c, &s);
    z[0] = R * c;
    z[1] = R * s;
}

/* 64x64 -> high 64 multiply; index draw i = floor(r * M / 2^64) */
static uint64_t mulhi64(uint64_t a, uint64_t b)
{
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
}

/* The two archive rows of a block-step from the two words of its first Philox block (0-based):
 * i1 ~ U{0..M-1}, i2 ~ U of the other M-1 -- the O(1) equivalent of collect(1:M) / rand / deleteat! / rand
 * (demcz.jl:176-179). */
static void draw_rows(uint64_t r1, uint64_t r2, int64_t M, uint64_t* i1, uint64_t* i2)
{
    *i1 = mulhi64(r1, (uint64_t)M);
    uint64_t j = mulhi64(r2, (uint64_t)(M - 1));
    *i2 = j + (j >= *i1 ? 1 : 0);
}

ORACLE_API void oracle_draw_rows(uint64_t seed, uint64_t chain, uint64_t blk, int64_t M, uint64_t out[2])
{
    uint64_t r1, r2;
    draw_block(seed, chain, blk, &r1, &r2);
    draw_rows(r1, r2, M, &out[0], &out[1]);
}

/* ------------------------------------------------------------------------------------------
 * Targets (the user log-densities the BASELINE configs exercise; NOT in the reference's
 * src/).
 * ------------------------------------------------------------------------------------------ */
enum { ORACLE_TARGET_MVNORMAL = 0, ORACLE_TARGET_ISO_QUAD = 1, ORACLE_TARGET_LINREG_SSE = 2 };

typedef struct oracle_problem {
    int64_t N;              /* chains handled by this call (local shard)                    */
    int64_t chain_id0;      /* global id of local chain 0: selects the Philox subsequence   */
    int32_t d;
    int32_t K;
    int64_t Mcap;           /* leading dimension (row capacity) of the column-major Z        */
    int32_t Nblocks;
    const int32_t* block_offsets;  /* CSR, Nblocks+1                                         */
    const int32_t* block_indices;  /* 0-based parameter indices                              */
    const double* eps_scale;       /* d                                                      */
    uint64_t seed;
    int32_t target_kind;
    const double* mu;       /* MVNORMAL, ISO_QUAD: d                                         */
    const double* W;        /* MVNORMAL: d x d column-major lower-triangular whitening L^-1  */
    double c0;              /* MVNORMAL: -0.5 (d log 2pi + logdet Sigma)                     */
    const double* design;   /* LINREG: nobs x d column-major                                 */
    const double* yobs;     /* LINREG: nobs                                                  */
    int64_t nobs;
} oracle_problem;

/* Whether the MvNormal sums are grouped by the blocks of the run: 2 <= Nblocks, and block b's index SET is the range
 * [o_b, o_{b+1}) with o the running sum of the block lengths (the blocks, in order, cut 0..d-1 into consecutive pieces;
 * the order of the indices inside a block does not matter).  Returns the number of groups (1: not grouped). */
static int mvn_groups(const oracle_problem* p, int32_t* off /* Nblocks + 1 */)
{
    const int d = p->d;
    if (p->Nblocks < 2 || p->Nblocks > d) return 1;
    if (p->block_offsets[p->Nblocks] != d) return 1;
    for (int b = 0; b < p->Nblocks; ++b) {
        const int lo = p->block_offsets[b], hi = p->block_offsets[b + 1];
        unsigned char seen[256] = {0};
        for (int t = lo; t < hi; ++t) {
            const int j = p->block_indices[t];
            if (j < lo || j >= hi || seen[j - lo]) return 1;
            seen[j - lo] = 1;
        }
        off[b] = lo;
    }
    off[p->Nblocks] = d;
    return p->Nblocks;
}

/* logpdf(MvNormal(mu, Sigma), x) as test/example_normpdf.jl:13-16 defines the target:
 * c0 - 0.5 * || W (x - mu) ||^2 with W = L^-1, Sigma = L L'.  Summation order is the spec.
 *   One group (Nblocks == 1, or blocks that are not consecutive ranges -- every case of rounds 1-3):
 *     y_i = W[i,0] r_0, then fma over j = 1..i; q = y_0^2 then fma over i; logp = fma(-0.5, q, c0).
 *   Grouped by the blocks (round 4; mvn_groups): with groups g = [o_g, o_{g+1}),
 *     P_ig = W[i,o_g] r_{o_g}, then fma over the further j of group g with j <= i      (the part of row i's dot product
 *                                                                                        that block g's coordinates feed)
 *     y_i  = P_i0 + P_i1 + ... + P_i,g(i)      plain additions, in group order, g(i) = the group of i
 *     Q_g  = y_{o_g}^2, then fma(y_i, y_i, Q_g) over the further i of group g
 *     q    = Q_0 + Q_1 + ...                   plain additions, in group order;   logp = fma(-0.5, q, c0).
 *   Why: update_blocks (src/demcz.jl:167-172) moves ONE block per block-step and re-evaluates the whole log-density
 *   (src/demcz.jl:189).  With the sums cut at the block boundaries a device kernel can keep P_ig and Q_g per chain and recompute
 *   only what the moved block feeds -- and get the very doubles this full re-evaluation gives.  A sum of d products has no
 *   canonical order in the reference (its arithmetic lives in Distributions / PDMats); with one group this is rounds 1-3's. */
static double target_logp(const oracle_problem* p, const double* x)
{
    const int d = p->d;
    if (p->target_kind == ORACLE_TARGET_MVNORMAL) {
        int32_t off[258];
        const int ng = (p->Nblocks + 1 <= 258) ? mvn_groups(p, off) : 1;
        if (ng > 1) {
            double q = 0.0;
            for (int g = 0; g < ng; ++g) {
                double Qg = 0.0;
                for (int i = off[g]; i < off[g + 1]; ++i) {
                    double y = 0.0;
                    for (int gb = 0; gb <= g; ++gb) {
                        const int jl = off[gb], jh = (off[gb + 1] - 1 < i) ? off[gb + 1] - 1 : i;
                        double P = p->W[i + (int64_t)d * jl] * (x[jl] - p->mu[jl]);
                        for (int j = jl + 1; j <= jh; ++j)
                            P = fma(p->W[i + (int64_t)d * j], x[j] - p->mu[j], P);
                        y = (gb == 0) ? P : y + P;
                    }
                    Qg = (i == off[g]) ? y * y : fma(y, y, Qg);
                }
                q = (g == 0) ? Qg : q + Qg;
            }
            return fma(-0.5, q, p->c0);
        }
        double q = 0.0;
        for (int i = 0; i < d; ++i) {
            double acc = p->W[i] * (x[0] - p->mu[0]);
            for (int j = 1; j <= i; ++j)
                acc = fma(p->W[i + (int64_t)d * j], x[j] - p->mu[j], acc);
            q = (i == 0) ? acc * acc : fma(acc, acc, q);
        }
        return fma(-0.5, q, p->c0);
    } else if (p->target_kind == ORACLE_TARGET_ISO_QUAD) {
        /* -sum((x - mu).^2), test/test_anneal.jl:10 */
        double q = 0.0;
        for (int i = 0; i < d; ++i) {
            double r = x[i] - p->mu[i];
            q = (i == 0) ? r * r : fma(r, r, q);
        }
        return -q;
    } else {
        /* -0.5 * sum((y - X*b).^2), test/example_linreg.jl:32.
         * Summation order of the spec: 16 interleaved partial sums (observation o goes to partial
         * o mod 16, accumulated in increasing o: first term r*r, then fma(r, r, partial)), combined
         * by the fixed tree (l, l+8), (l, l+4), (l, l+2), (0, 1).  A sum of 1000 squares has no
         * canonical order in the reference (Julia's sum() is itself a pairwise reduction); this one
         * lets 16 lanes / 16 accumulators work on one chain. */
        double part[16];
        for (int l = 0; l < 16; ++l) {
            double s = 0.0;
            int first = 1;
            for (int64_t o = l; o < p->nobs; o += 16) {
                double acc = p->design[o] * x[0];
                for (int j = 1; j < d; ++j)
                    acc = fma(p->design[o + p->nobs * j], x[j], acc);
                double r = p->yobs[o] - acc;
                s = first ? r * r : fma(r, r, s);
                first = 0;
            }
            part[l] = s;
        }
        for (int h = 8; h >= 1; h >>= 1)
            for (int l = 0; l < h; ++l) part[l] = part[l] + part[l + h];
        return -0.5 * part[0];
    }
}

ORACLE_API void oracle_logp(const oracle_problem* p, const double* X, int64_t ldX, int64_t n, double* out)
{
    double x[256];
    for (int64_t c = 0; c < n; ++c) {
        for (int j = 0; j < p->d; ++j) x[j] = X[c + ldX * j];
        out[c] = target_logp(p, x);
    }
}

/* Number of Philox blocks one block-step of block length b consumes:
 * 1 (two index words) + ceil(nn/2) normal pairs + 1 (accept uniform), nn = (b == 1 ? 1 : b). */
static int64_t blockstep_nblk(int b)
{
    int nn = (b == 1) ? 1 : b;
    return 1 + (nn + 1) / 2 + 1;
}

ORACLE_API int64_t oracle_blocks_per_generation(const oracle_problem* p)
{
    int64_t s = 0;
    for (int ib = 0; ib < p->Nblocks; ++ib)
        s += blockstep_nblk(p->block_offsets[ib + 1] - p->block_offsets[ib]);
    return s;
}

/* One block-step: src/demcz.jl:174-195 (sampler) / src/demcz_anneal.jl:149-170 (annealer).
 *   draw order per block-step (reference: i1, i2, the normals, the accept uniform):
 *     block 0          : i1 = floor(r1 M / 2^64); j = floor(r2 (M-1) / 2^64); i2 = j + (j >= i1)
 *                        (O(1) equivalent of collect(1:M)/deleteat!, demcz.jl:176-179)
 *     block 1..npairs  : Box-Muller pairs
 *     block 1+npairs   : logu = dm_log(u_open(r1))
 *   blocklen == 1: gamma unscaled and ONE scalar normal (demcz.jl:183-184);
 *   otherwise gamma / sqrt(2 b) and b normals (demcz.jl:186).
 *   accept: log(u) < lp - logp (strict; NaN rejects), tempered: < (lp - logp) / T.
 * Returns 1 if accepted. `dbg` (optional, 8+ doubles + normals) receives the draws. */
static int block_step(const oracle_problem* p, const double* Z, int64_t M, uint64_t chain, uint64_t blk0,
                      int ib, double gamma, const double* temperature, double* x, double* logp,
                      double* dbg)
{
    const int d = p->d;
    const int32_t* blk = p->block_indices + p->block_offsets[ib];
    const int b = p->block_offsets[ib + 1] - p->block_offsets[ib];
    const int nn = (b == 1) ? 1 : b;
    const int npairs = (nn + 1) / 2;
    uint64_t r1, r2;
    draw_block(p->seed, chain, blk0, &r1, &r2);
    uint64_t i1, i2;
    draw_rows(r1, r2, M, &i1, &i2);
    double zn[256];
    for (int pr = 0; pr < npairs; ++pr) {
        draw_block(p->seed, chain, blk0 + 1 + (uint64_t)pr, &r1, &r2);
        oracle_normal_pair(r1, r2, zn + 2 * pr);
    }
    draw_block(p->seed, chain, blk0 + 1 + (uint64_t)npairs, &r1, &r2);
    double logu = oracle_dm_log(u_open(r1));

    double scale = (b == 1) ? gamma : gamma / sqrt((double)(2 * b));
    double xp[256];
    for (int t = 0; t < d; ++t) xp[t] = x[t];
    for (int t = 0; t < b; ++t) {
        int pi = blk[t];
        double diff = Z[(int64_t)i1 + p->Mcap * pi] - Z[(int64_t)i2 + p->Mcap * pi];
        double zt = (b == 1) ? zn[0] : zn[t];
        double t1 = scale * diff;
        double t2 = p->eps_scale[pi] * zt;
        double delta = t1 + t2;
        xp[pi] = x[pi] + delta;
    }
    double lp = target_logp(p, xp);
    double dlt = lp - *logp;
    if (temperature) dlt = dlt / *temperature;
    int acc = (logu < dlt) ? 1 : 0;
    if (dbg) {
        dbg[0] = (double)i1; dbg[1] = (double)i2; dbg[2] = logu; dbg[3] = lp; dbg[4] = (double)acc;
        for (int t = 0; t < nn; ++t) dbg[5 + t] = zn[t];
        for (int t = 0; t < d; ++t) dbg[5 + nn + t] = xp[t];
    }
    if (acc) {
        for (int t = 0; t < d; ++t) x[t] = xp[t];
        *logp = lp;
    }
    return acc;
}

/* Single block-step on one chain, exposing every intermediate (fixture tier (i)). */
ORACLE_API int oracle_block_step(const oracle_problem* p, const double* Z, int64_t M, int64_t chain_local,
                                 int64_t g, int ib, double gamma, const double* temperature,
                                 double* x, double* logp, double* dbg)
{
    int64_t S = oracle_blocks_per_generation(p);
    int64_t off = 0;
    for (int t = 0; t < ib; ++t) off += blockstep_nblk(p->block_offsets[t + 1] - p->block_offsets[t]);
    uint64_t blk0 = (uint64_t)(g - 1) * (uint64_t)S + (uint64_t)off;
    return block_step(p, Z, M, (uint64_t)(p->chain_id0 + chain_local), blk0, ib, gamma, temperature, x, logp, dbg);
}

enum { ORACLE_SCHED_SYNCHRONOUS = 0, ORACLE_SCHED_SEQUENTIAL = 1 };

/* Generations g_from..g_to (1-based, inclusive) for all N local chains.
 *   src/demcz.jl:30-33 (generation loop, chains in order), :80-93 (runchain!: history write,
 *   Xcurrent/log_objcurrent update, Z append when g % K == 0), :167-172 (update_blocks).
 * schedule SEQUENTIAL is the reference's Gauss-Seidel order: chain ic appends its row
 *   before chain ic+1 proposes (Q2).  SYNCHRONOUS (what the GPU does): every chain of a
 *   generation proposes against the same M, then rows M..M+N-1 are appended in chain order.
 * do_append == 0 leaves Z/M untouched (sharded runs: the caller gathers and appends).
 * Layouts: X (N x d, ld N), Z (Mcap x d, ld Mcap), chain_out (N x d x G), logobj_out (N x G),
 *   all column-major like the Julia arrays (DEMC.jl:10-15).  changed_out[g - g_from] counts
 *   chains whose log_obj differs from the previous generation's (the event demcz.jl:42 and
 *   demcz_anneal.jl:50 count through diff(log_obj) .!= 0).
 * temperature: NULL (sampler) or one value per generation (annealer, demcz_anneal.jl:69). */
ORACLE_API int oracle_demcz_run(const oracle_problem* p, double* X, double* logp, double* Z, int64_t* M,
                                int64_t g_from, int64_t g_to, double gamma, const double* temperature,
                                double* chain_out, double* logobj_out, int64_t* changed_out,
                                int schedule, int do_append, int64_t rng_offset)
{
    const int64_t N = p->N;
    const int d = p->d;
    if (d > 256 || *M < 2) return 1;
    const int64_t S = oracle_blocks_per_generation(p);
    double x[256];
    for (int64_t g = g_from; g <= g_to; ++g) {
        const int64_t gi = g - g_from;
        const double* T = temperature ? &temperature[gi] : NULL;
        const int64_t Mgen = *M;
        int64_t changed = 0;
        for (int64_t c = 0; c < N; ++c) {
            for (int t = 0; t < d; ++t) x[t] = X[c + N * t];
            double lp = logp[c];
            const double lp_before = lp;
            const int64_t Mvis = (schedule == ORACLE_SCHED_SEQUENTIAL) ? *M : Mgen;
            /* rng_offset: generations a previous run already drew from every chain's stream (resume) */
            uint64_t blk0 = (uint64_t)(g + rng_offset - 1) * (uint64_t)S;
            for (int ib = 0; ib < p->Nblocks; ++ib) {
                block_step(p, Z, Mvis, (uint64_t)(p->chain_id0 + c), blk0, ib, gamma, T, x, &lp, NULL);
                blk0 += (uint64_t)blockstep_nblk(p->block_offsets[ib + 1] - p->block_offsets[ib]);
            }
            for (int t = 0; t < d; ++t) X[c + N * t] = x[t];
            logp[c] = lp;
            if ((lp - lp_before) != 0.0) ++changed;      /* diff(log_obj) .!= 0: a NaN difference counts (demcz.jl:42) */
            if (chain_out)
                for (int t = 0; t < d; ++t) chain_out[c + N * (t + (int64_t)d * gi)] = x[t];
            if (logobj_out) logobj_out[c + N * gi] = lp;
            if (do_append && schedule == ORACLE_SCHED_SEQUENTIAL && (g % p->K) == 0) {
                if (*M >= p->Mcap) return 2;
                for (int t = 0; t < d; ++t) Z[*M + p->Mcap * t] = x[t];
                *M += 1;
            }
        }
        if (changed_out) changed_out[gi] = changed;
        if (do_append && schedule == ORACLE_SCHED_SYNCHRONOUS && (g % p->K) == 0) {
            if (*M + N > p->Mcap) return 2;
            for (int64_t c = 0; c < N; ++c)
                for (int t = 0; t < d; ++t) Z[*M + c + p->Mcap * t] = X[c + N * t];
            *M += N;
        }
    }
    return 0;
}

/* Synchronous schedule with the chains of a generation spread over the host's cores (OpenMP).
 * Same arithmetic and the same per-chain streams as oracle_demcz_run(..., ORACLE_SCHED_SYNCHRONOUS, ...):
 * the result is bit-identical for any thread count (tests/test_oracle_sampler.py).  Exists for the
 * "CPU-omp" baseline row; without -fopenmp it is the serial loop. */
ORACLE_API int oracle_demcz_run_omp(const oracle_problem* p, double* X, double* logp, double* Z, int64_t* M,
                                    int64_t g_from, int64_t g_to, double gamma, const double* temperature,
                                    double* chain_out, double* logobj_out, int64_t* changed_out,
                                    int do_append, int64_t rng_offset, int nthreads)
{
    const int64_t N = p->N;
    const int d = p->d;
    if (d > 256 || *M < 2) return 1;
    const int64_t S = oracle_blocks_per_generation(p);
    (void)nthreads;
    for (int64_t g = g_from; g <= g_to; ++g) {
        const int64_t gi = g - g_from;
        const double* T = temperature ? &temperature[gi] : NULL;
        const int64_t Mgen = *M;
        int64_t changed = 0;
#pragma omp parallel for schedule(static) reduction(+ : changed) num_threads(nthreads > 0 ? nthreads : 1)
        for (int64_t c = 0; c < N; ++c) {
            double x[256];
            for (int t = 0; t < d; ++t) x[t] = X[c + N * t];
            double lp = logp[c];
            const double lp_before = lp;
            uint64_t blk0 = (uint64_t)(g + rng_offset - 1) * (uint64_t)S;
            for (int ib = 0; ib < p->Nblocks; ++ib) {
                block_step(p, Z, Mgen, (uint64_t)(p->chain_id0 + c), blk0, ib, gamma, T, x, &lp, NULL);
                blk0 += (uint64_t)blockstep_nblk(p->block_offsets[ib + 1] - p->block_offsets[ib]);
            }
            for (int t = 0; t < d; ++t) X[c + N * t] = x[t];
            logp[c] = lp;
            if ((lp - lp_before) != 0.0) ++changed;      /* diff(log_obj) .!= 0: a NaN difference counts (demcz.jl:42) */
            if (chain_out)
                for (int t = 0; t < d; ++t) chain_out[c + N * (t + (int64_t)d * gi)] = x[t];
            if (logobj_out) logobj_out[c + N * gi] = lp;
        }
        if (changed_out) changed_out[gi] = changed;
        if (do_append && (g % p->K) == 0) {
            if (*M + N > p->Mcap) return 2;
            for (int64_t c = 0; c < N; ++c)
                for (int t = 0; t < d; ++t) Z[*M + c + p->Mcap * t] = X[c + N * t];
            *M += N;
        }
    }
    return 0;
}

/* Same schedule, but with the reference's O(M) index draw cost emulated
 * (collect(1:M) + deleteat!, demcz.jl:176-178): materialise and shift an M-vector per
 * block-step.  Used only for the "faithful-cost" CPU baseline row; results are identical. */
ORACLE_API int64_t oracle_faithful_index_cost(int64_t M, int64_t i1)
{
    int64_t* set = (int64_t*)malloc((size_t)M * sizeof(int64_t));
    if (!set) return -1;
    for (int64_t i = 0; i < M; ++i) set[i] = i + 1;
    memmove(set + i1, set + i1 + 1, (size_t)(M - 1 - i1) * sizeof(int64_t));
    int64_t r = set[M / 2];
    free(set);
    return r;
}

/* T(ig) = T0 (TN/T0)^(ig/Ng), src/demcz_anneal.jl:1-3. */
ORACLE_API double oracle_tempbaseline(int64_t ig, int64_t Ng, double T0, double TN)
{
    return T0 * pow(TN / T0, (double)ig / (double)Ng);
}

/* Split-chain Gelman-Rubin R-hat, src/utils.jl:2-20.
 * chain: N x d x G column-major window; n = floor(G/2), m = 2N; halves 1:n and n+1:2n
 * (an odd window drops its last sample, utils.jl:4-8). */
ORACLE_API int oracle_rhat_gelman(const double* chain, int64_t N, int64_t G, int32_t d, double* rhat)
{
    const int64_t n = G / 2;
    const int64_t m = 2 * N;
    if (n < 2 || m < 2) return 1;
    double* mean_j = (double*)malloc((size_t)m * sizeof(double));
    if (!mean_j) return 2;
    for (int p = 0; p < d; ++p) {
        double grand = 0.0;
        for (int64_t h = 0; h < 2; ++h)
            for (int64_t c = 0; c < N; ++c) {
                double s = 0.0;
                for (int64_t t = 0; t < n; ++t) s += chain[c + N * (p + (int64_t)d * (h * n + t))];
                mean_j[h * N + c] = s / (double)n;
                grand += s;
            }
        grand /= (double)(m * n);                       /* utils.jl:10 */
        double B = 0.0, W = 0.0;
        for (int64_t h = 0; h < 2; ++h)
            for (int64_t c = 0; c < N; ++c) {
                double mj = mean_j[h * N + c];
                B += (mj - grand) * (mj - grand);       /* utils.jl:13 */
                double s2 = 0.0;
                for (int64_t t = 0; t < n; ++t) {
                    double v = chain[c + N * (p + (int64_t)d * (h * n + t))] - mj;
                    s2 += v * v;
                }
                W += s2 / (double)(n - 1);              /* utils.jl:14 */
            }
        B *= (double)n / (double)(m - 1);
        W /= (double)m;                                 /* utils.jl:15 */
        double varhat = (double)(n - 1) / (double)n * W + B / (double)n;   /* utils.jl:16 */
        rhat[p] = sqrt(varhat / W);                     /* utils.jl:18 */
    }
    free(mean_j);
    return 0;
}

/* Per-chain count of generations whose log_obj differs from the previous column:
 * sum(diff(log_obj, dims=2) .!= 0, dims=2), src/utils.jl:61 and the intent of demcz.jl:42.
 * log_obj: N x G column-major. */
ORACLE_API void oracle_changed_per_chain(const double* log_obj, int64_t N, int64_t G, int64_t* out)
{
    for (int64_t c = 0; c < N; ++c) {
        int64_t k = 0;
        for (int64_t g = 1; g < G; ++g)
            if ((log_obj[c + N * g] - log_obj[c + N * (g - 1)]) != 0.0) ++k;     /* diff(.) .!= 0, utils.jl:61 */
        out[c] = k;
    }
}

/* mean_cov_chain, src/utils.jl:96-111 via flatten_chain :22-32: mean over all N*G draws and
 * the 1/(N G) population covariance.  cov: d x d column-major. */
ORACLE_API void oracle_mean_cov_chain(const double* chain, int64_t N, int64_t G, int32_t d, double* mean, double* cov)
{
    const double cnt = (double)(N * G);
    for (int p = 0; p < d; ++p) {
        double s = 0.0;
        for (int64_t g = 0; g < G; ++g)
            for (int64_t c = 0; c < N; ++c) s += chain[c + N * (p + (int64_t)d * g)];
        mean[p] = s / cnt;
    }
    for (int p = 0; p < d; ++p)
        for (int q = 0; q < d; ++q) {
            double s = 0.0;
            for (int64_t g = 0; g < G; ++g)
                for (int64_t c = 0; c < N; ++c)
                    s += (chain[c + N * (p + (int64_t)d * g)] - mean[p]) * (chain[c + N * (q + (int64_t)d * g)] - mean[q]);
            cov[p + d * q] = s / cnt;
        }
}
