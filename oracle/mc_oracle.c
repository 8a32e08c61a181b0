/*
 * mc_oracle.c -- TEST INFRASTRUCTURE ONLY (see mc_oracle.h for the contract and the
 * parity-pinning statement).  Instantiates mc_oracle_impl.h for f32 and f64 and holds the
 * precision-independent pieces: Philox4x32-10 and the fp64 closing formulas.
 */
#define _GNU_SOURCE
#include <math.h>
#include <string.h>
#include <stdint.h>
#include <stdlib.h>

#include "mc_oracle.h"

/* ------------------------------------------------------------------------------------- */
/* Philox4x32-10, as published (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as
 * easy as 1, 2, 3", SC'11; Random123 v1.09 philox.h).  rocRAND / cuRAND ship the same
 * generator (the reference draws from cuRAND: dp/MonteCarloKernel.cu:68,78,250,289; its
 * XORWOW stream is not reproduced -- BASELINE.json north_star allows Philox).
 * Pinned by the Random123 known-answer vectors in tests/golden/philox_kat.json.           */
/* ------------------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u; /* round multipliers */
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u; /* Weyl key increments */
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += W0, k1 += W1;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

/* ------------------------------------------------------------------------------------- */
/* XORWOW (Marsaglia, "Xorshift RNGs", 2003: a 160-bit xorshift plus a Weyl sequence) -- the generator behind the
 * reference's curand_init / curand_normal (dp/MonteCarloKernel.cu:285-290,68,78,250) and rocRAND's default device
 * generator.  The product offers it as a second generator; this is its oracle twin.
 *   state: x[5] (xorshift), d (Weyl);  next(): t = x0 ^ (x0 >> 2); shift the words down;
 *          x4 = (x4 ^ (x4 << 4)) ^ (t ^ (t << 1)); d += 362437; return d + x4
 *   init(seed, subsequence): rocRAND's seeding (rocrand_xorwow.h: fixed start words scrambled with two prime
 *          multiples of the seed halves), then a jump of subsequence * 2^67 steps of the xorshift part (the Weyl word
 *          is unchanged: 2^67 is a multiple of 2^32).  The jump is linear algebra over GF(2): the matrices
 *          A^(2^67 * 2^i) are computed HERE by repeated squaring of the one-step matrix -- no table is copied from
 *          rocRAND.  Pinned: tests/test_rocrand_xcheck.py compares these words with rocRAND's own host-callable
 *          engine (rocrand_init(seed, subsequence, 0) + rocrand()).  This reproduces rocRAND's sequence; cuRAND
 *          7.5's seeding is not public in the reference tree: "parity unpinned" for that one.                       */
/* ------------------------------------------------------------------------------------- */
typedef struct { uint32_t w[5]; } xw_vec;
typedef struct { xw_vec col[160]; } xw_mat;   /* column c = image of basis bit c (word c / 32, bit c % 32) */

static xw_vec xw_step_linear(xw_vec v)
{
    const uint32_t t = v.w[0] ^ (v.w[0] >> 2);
    xw_vec r;
    r.w[0] = v.w[1], r.w[1] = v.w[2], r.w[2] = v.w[3], r.w[3] = v.w[4];
    r.w[4] = (v.w[4] ^ (v.w[4] << 4)) ^ (t ^ (t << 1));
    return r;
}
static xw_vec xw_mul_vec(const xw_mat *m, xw_vec v)
{
    xw_vec r = {{0, 0, 0, 0, 0}};
    for (int c = 0; c < 160; c++)
        if ((v.w[c / 32] >> (c % 32)) & 1u)
            for (int k = 0; k < 5; k++)
                r.w[k] ^= m->col[c].w[k];
    return r;
}
static void xw_mul_mat(xw_mat *out, const xw_mat *b, const xw_mat *a)   /* out = b * a (a applied first) */
{
    xw_mat r;
    for (int c = 0; c < 160; c++)
        r.col[c] = xw_mul_vec(b, a->col[c]);
    *out = r;
}

#define XW_SUBSEQ_BITS 48
static xw_mat xw_jump[XW_SUBSEQ_BITS];   /* xw_jump[i] = A^(2^67 * 2^i) */
static int xw_ready;

static void xw_prepare(void)
{
    if (xw_ready)
        return;
    xw_mat a;
    for (int c = 0; c < 160; c++) {
        xw_vec e = {{0, 0, 0, 0, 0}};
        e.w[c / 32] = 1u << (c % 32);
        a.col[c] = xw_step_linear(e);
    }
    for (int i = 0; i < 67; i++)
        xw_mul_mat(&a, &a, &a);
    xw_jump[0] = a;
    for (int i = 1; i < XW_SUBSEQ_BITS; i++)
        xw_mul_mat(&xw_jump[i], &xw_jump[i - 1], &xw_jump[i - 1]);
    xw_ready = 1;
}

/* state[0..4] = x, state[5] = d */
void orc_xorwow_init(uint64_t seed, uint64_t subsequence, uint32_t state[6])
{
    xw_prepare();
    xw_vec x = {{123456789u, 362436069u, 521288629u, 88675123u, 5783321u}};
    uint32_t d = 6615241u;
    const uint32_t s0 = (uint32_t)seed ^ 0x2c7f967fu, s1 = (uint32_t)(seed >> 32) ^ 0xa03697cbu;
    const uint32_t t0 = 1228688033u * s0, t1 = 2073658381u * s1;
    x.w[0] += t0, x.w[1] ^= t0, x.w[2] += t1, x.w[3] ^= t1, x.w[4] += t0;
    d += t1 + t0;
    for (int i = 0; i < XW_SUBSEQ_BITS; i++)
        if ((subsequence >> i) & 1u)
            x = xw_mul_vec(&xw_jump[i], x);
    for (int k = 0; k < 5; k++)
        state[k] = x.w[k];
    state[5] = d;
}

uint32_t orc_xorwow_next(uint32_t state[6])
{
    const uint32_t t = state[0] ^ (state[0] >> 2);
    state[0] = state[1], state[1] = state[2], state[2] = state[3], state[3] = state[4];
    state[4] = (state[4] ^ (state[4] << 4)) ^ (t ^ (t << 1));
    state[5] += 362437u;
    return state[5] + state[4];
}

/* The reference's per-thread normal stream under its launch geometry (dp/MonteCarloKernel.cu:285-290: curand_init(seed =
 * blockIdx.x + gridDim.x, subsequence = threadIdx.x, offset 0); :68,78,250 curand_normal): the first `count` normals of
 * thread `thread` of block `block` of a launch of `num_blocks` blocks.  cuRAND is not in the image; this follows its AMD
 * counterpart, which a HIP build of the reference calls: rocrand_normal(rocrand_state_xorwow *) -- two words per
 * Box-Muller pair, sine member first, cosine member kept for the next call -- and box_muller(x, y) of
 * rocrand_normal.h (host branch: sinf / cosf), expression for expression.  Pinned against rocRAND's own host-callable
 * engine by tests/test_rocrand_xcheck.py. */
void orc_grid_normals(uint32_t num_blocks, uint32_t block, uint32_t thread, uint32_t count, float *out)
{
    uint32_t st[6];
    orc_xorwow_init((uint64_t)block + num_blocks, thread, st);
    for (uint32_t k = 0; k < count; k += 2) {
        const unsigned int x = orc_xorwow_next(st), y = orc_xorwow_next(st);
        const float u = 2.3283064e-10f + (x * 2.3283064e-10f);
        const float v = 1.46291807e-09f + (y * 1.46291807e-09f);
        const float s = sqrtf(-2.0f * logf(u));
        out[k] = sinf(v) * s;
        if (k + 1 < count)
            out[k + 1] = cosf(v) * s;
    }
}

/* the jump matrices themselves, for the test that compares the product's (computed the same way, independently) */
void orc_xorwow_jump_column(int i, int c, uint32_t out[5])
{
    xw_prepare();
    for (int k = 0; k < 5; k++)
        out[k] = xw_jump[i].col[c].w[k];
}

/* XORWOW as the normal source of the orc_dev_* family.  The product runs one XORWOW sequence per lane of the launch
 * (lane l = subsequence base + l, like the reference's one curandState per thread, dp/MonteCarloKernel.cu:285-290);
 * lane l prices units unit0 + l, unit0 + l + lanes, ... and draws four words per Philox-block-equivalent, in the
 * order the kernels ask for them.  Between orc_xorwow_begin and orc_xorwow_end the normals of (unit, block) therefore
 * come from the owning lane's NEXT four words: callers must ask for every block exactly once, units ascending --
 * which is how the orc_dev_* loops run. */
static struct { int on; uint64_t unit0; uint32_t lanes; uint32_t *state; } xw_mode;

void orc_xorwow_begin(uint64_t seed, uint64_t subsequence_base, uint32_t lanes, uint64_t unit0)
{
    xw_mode.on = 1, xw_mode.unit0 = unit0, xw_mode.lanes = lanes;
    xw_mode.state = (uint32_t *)malloc(sizeof(uint32_t) * 6 * (size_t)lanes);
    for (uint32_t l = 0; l < lanes; l++)
        orc_xorwow_init(seed, subsequence_base + l, xw_mode.state + 6 * (size_t)l);
}
void orc_xorwow_end(void)
{
    free(xw_mode.state);
    xw_mode.on = 0, xw_mode.state = NULL;
}
/* four words for (unit, block): Philox by default, the lane's XORWOW sequence in XORWOW mode */
static void orc_block_words(uint64_t seed, uint32_t domain, uint64_t unit, uint32_t block, uint32_t x[4])
{
    if (xw_mode.on) {
        uint32_t *st = xw_mode.state + 6 * (size_t)((unit - xw_mode.unit0) % xw_mode.lanes);
        for (int k = 0; k < 4; k++)
            x[k] = orc_xorwow_next(st);
        return;
    }
    uint32_t ctr[4] = {(uint32_t)(unit >> 32), (uint32_t)unit, block, domain};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_philox4x32_10(ctr, key, x);
}
static int orc_xorwow_active(void) { return xw_mode.on; }

/* fp64 family: 0 = native fp64 normals (2 per block), 1 = four fp32 normals per block widened to double (the reference's
 * dp arithmetic; twin of the product's MC_NORMALS_F32).  See mc_oracle_impl.h: orc_dev_normals. */
static int orc_normals_f32_mode;
void orc_set_normals_f32(int on) { orc_normals_f32_mode = on != 0; }

/* price = discount * sum/n;  s^2 = (n sum2 - sum^2) / (n (n-1));  CI = 1.96 s / sqrt(n).
 * dp/MonteCarloHost.c:220-228, dp/MonteCarloKernel.cu:420-423 (and :466-468 for CVA). */
void orc_closing(double sum, double sum2, long long n, double discount, double *expected,
                 double *confidence)
{
    double dn = (double)n;
    *expected = discount * (sum / dn);
    double dev = sqrt((dn * sum2 - sum * sum) / (dn * (double)(n - 1)));
    *confidence = 1.96 * dev / sqrt(dn);
}

/* ---- f32 instantiation ---- */
#define REAL float
#define X f32
#define SQRT_R sqrtf
#define LOG_R logf
#define EXP_R expf
#define ORC_IS_F32 1
#define ORC_NPB 4
#include "mc_oracle_impl.h"
#undef REAL
#undef X
#undef SQRT_R
#undef LOG_R
#undef EXP_R
#undef ORC_IS_F32
#undef ORC_NPB

/* ---- f64 instantiation ---- */
#define REAL double
#define X f64
#define SQRT_R sqrt
#define LOG_R log
#define EXP_R exp
#define ORC_IS_F32 0
#define ORC_NPB 8
#include "mc_oracle_impl.h"
