/*
 * mc_oracle_impl.h -- TEST INFRASTRUCTURE ONLY; precision-generic body, included twice by
 * mc_oracle.c with
 *     REAL   float | double          X      f32 | f64
 *     SQRT_R sqrtf | sqrt            LOG_R  logf | log          EXP_R expf | exp
 *
 * Reference line numbers are for double_precision/ ("dp/"); the single_precision/ twin
 * ("sp/") is the same text with float/sqrtf/logf/expf (SURVEY.md conventions).  Where the two
 * differ in more than the type, both are cited.
 *
 * NOTE ON LITERALS: the reference mixes double literals (0.5, 1.0, -2.0, 2.0*M_PI, 1.96) into
 * float expressions in sp/.  C's usual arithmetic conversions make those sub-expressions
 * double, and the bits depend on it.  The host family below therefore spells every
 * promotion as an explicit cast, so that one body reproduces dp/ and sp/ bit for bit.
 */

#define ORC_CAT_(a, b) a##_##b
#define ORC_CAT(a, b) ORC_CAT_(a, b)
#define FN(name) ORC_CAT(name, X)

/* ===================================================================================== */
/*  Deterministic pieces                                                                  */
/* ===================================================================================== */

/* Hastings / Abramowitz-Stegun 26.2.17 normal CDF.  dp/MonteCarloHost.c:124-136 (identical
 * constants on the device, dp/MonteCarloKernel.cu:110-123). */
REAL FN(orc_cnd)(REAL d)
{
    const REAL a1 = (REAL)0.31938153, a2 = (REAL)-0.356563782, a3 = (REAL)1.781477937;
    const REAL a4 = (REAL)-1.821255978, a5 = (REAL)1.330274429;
    const REAL inv_sqrt_2pi = (REAL)0.39894228040143267793994605993438;
    /* `1.0 / (1.0 + 0.2316419 * fabs(d))` is a double expression in both precisions */
    REAL kk = (REAL)(1.0 / (1.0 + 0.2316419 * fabs((double)d)));
    REAL poly = kk * (a1 + kk * (a2 + kk * (a3 + kk * (a4 + kk * a5))));
    /* `- 0.5 * d * d` is double; the exponential is taken in REAL */
    REAL tail = inv_sqrt_2pi * EXP_R((REAL)(-0.5 * (double)d * (double)d)) * poly;
    if (d > 0)
        tail = (REAL)(1.0 - (double)tail);
    return tail;
}

/* Closed-form Black-Scholes call.  dp/MonteCarloHost.c:139-143. */
REAL FN(orc_bs_call)(REAL s, REAL k, REAL r, REAL v, REAL t)
{
    REAL sqrt_t = SQRT_R(t);
    double num = (double)LOG_R(s / k) + ((double)r + 0.5 * (double)v * (double)v) * (double)t;
    REAL d1 = (REAL)(num / (double)(v * sqrt_t));
    REAL d2 = d1 - v * sqrt_t;
    return s * FN(orc_cnd)(d1) - k * EXP_R(-r * t) * FN(orc_cnd)(d2);
}

/* Cholesky, column by column, with the reference's zero-pivot rule (a column whose pivot is
 * not positive is left zero).  dp/MonteCarloHost.c:90-105.  Row-major n x n, runtime n. */
void FN(orc_chol)(int n, const REAL *c, REAL *a)
{
    REAL *work = (REAL *)malloc(sizeof(REAL) * (size_t)n);
    for (int col = 0; col < n; col++) {
        for (int row = 0; row < n; row++) {
            a[row * n + col] = 0;
            if (row < col)
                continue;
            work[row] = c[row * n + col];
            for (int q = 0; q < col; q++)
                work[row] -= a[col * n + q] * a[row * n + q];
            if (work[col] > 0)
                a[row * n + col] = work[row] / SQRT_R(work[col]);
        }
    }
    free(work);
}

/* Covariance input: what a caller of the reference would do by hand before Chol when it holds a
 * covariance matrix instead of (vols, correlation) -- v_a = sqrt(cov_aa), corr_ab = cov_ab / (v_a v_b) from the
 * lower triangle, unit diagonal -- followed by the reference's Chol (dp/basketOpt.cu:96-99, dp/MonteCarloHost.c:90-105).
 * Twin of mc_factor_from_cov_*; the Chol step is pinned by tests/golden/ref_chol.json and ref_cov.json. */
int FN(orc_factor_from_cov)(int n, const REAL *cov, REAL *v, REAL *corr, REAL *a)
{
    for (int i = 0; i < n; i++) {
        if (!(cov[i * n + i] > 0))
            return -1;
        v[i] = SQRT_R(cov[i * n + i]);
    }
    for (int i = 0; i < n; i++) {
        corr[i * n + i] = 1;
        for (int j = 0; j < i; j++)
            corr[i * n + j] = corr[j * n + i] = cov[i * n + j] / (v[i] * v[j]);
    }
    FN(orc_chol)(n, corr, a);
    int bad = 0;
    for (int i = 0; i < n; i++)
        bad += !(a[i * n + i] > 0);
    return bad;
}

/* ===================================================================================== */
/*  Reference CPU random stream: glibc rand(), cosine-branch Box-Muller                    */
/* ===================================================================================== */

/* dp/MonteCarloHost.c:111-114 */
static REAL FN(host_rand_between)(REAL lo, REAL hi)
{
    REAL x = (REAL)rand() / (REAL)(RAND_MAX);
    return hi * x + (REAL)(1.0f - x) * lo; /* 1.0f - x: float in sp, promoted in dp */
}

/* dp/MonteCarloHost.c:117-121; sp/:118-122 keeps `cos` and `2.0 * M_PI` in double */
static REAL FN(host_gaussian)(REAL mu, REAL sigma)
{
    REAL x = FN(host_rand_between)(0, 1);
    REAL y = FN(host_rand_between)(0, 1);
    REAL radius = SQRT_R((REAL)(-2.0 * (double)LOG_R(x)));
    double c = cos(2.0 * M_PI * (double)y);
    return (REAL)((double)mu + (double)sigma * ((double)radius * c));
}

void FN(orc_host_uniforms)(unsigned seed, int count, REAL *out)
{
    srand(seed);
    for (int i = 0; i < count; i++)
        out[i] = FN(host_rand_between)(0, 1);
}

void FN(orc_host_gaussians)(unsigned seed, int count, REAL *out)
{
    srand(seed);
    for (int i = 0; i < count; i++)
        out[i] = FN(host_gaussian)(0, 1);
}

/* closing, reference arithmetic: dp/MonteCarloHost.c:220-228 (prices, discounted) and
 * :270-275 (CVA, not discounted).  sp/ does all of it in float with sqrtf/expf. */
static void FN(host_close)(REAL sum, REAL sum2, int paths, int discounted, REAL r, REAL t,
                           orc_result *out)
{
    REAL mean = sum / (REAL)paths;
    REAL price = discounted ? EXP_R(-r * t) * mean : mean;
    REAL dev = SQRT_R(((REAL)paths * sum2 - sum * sum) / ((REAL)paths * (REAL)(paths - 1)));
    /* `1.96 * dev / sqrt(paths)`: double in dp (sqrt of an int), `1.96 * dev / sqrtf(paths)`
     * in sp = double * float / float -> double, stored into a REAL field */
    REAL conf = (REAL)(1.96 * (double)dev / (double)SQRT_R((REAL)paths));
    out->expected = (double)price;
    out->confidence = (double)conf;
    out->sum = (double)sum;
    out->sum2 = (double)sum2;
    out->n = paths;
}

/* Vanilla: dp/MonteCarloHost.c:170-173 (payoff) inside :185-199 (loop).  `payoffs` (optional, test tap): every path's
 * payoff as the reference forms it -- the arithmetic is untouched, the E/CI stay bit-pinned to the compiled reference. */
void FN(orc_host_vanilla_paths)(REAL s, REAL k, REAL r, REAL v, REAL t, int paths, unsigned seed,
                                REAL *payoffs, orc_result *out)
{
    REAL sum = 0, sum2 = 0;
    srand(seed);
    for (int i = 0; i < paths; i++) {
        REAL g = FN(host_gaussian)(0, 1);
        double drift = ((double)r - 0.5 * (double)v * (double)v) * (double)t;
        REAL diffusion = g * SQRT_R(t) * v;
        REAL value = s * EXP_R((REAL)(drift + (double)diffusion)) - k;
        REAL payoff = value > 0 ? value : 0;
        if (payoffs)
            payoffs[i] = payoff;
        sum += payoff;
        sum2 += payoff * payoff;
    }
    FN(host_close)(sum, sum2, paths, 1, r, t, out);
}
void FN(orc_host_vanilla)(REAL s, REAL k, REAL r, REAL v, REAL t, int paths, unsigned seed,
                          orc_result *out)
{
    FN(orc_host_vanilla_paths)(s, k, r, v, t, paths, seed, NULL, out);
}

/* Basket: dp/MonteCarloHost.c:150-161 (correlated normals: n draws, FULL n x n product,
 * + drift), :176-183 (terminal spots) and :200-218 (loop).
 *   vol_in_diffusion = 1 : si = v[i]*g[i]*sqrt(t)   (sp/MonteCarloHost.c:182, and the device)
 *   vol_in_diffusion = 0 : si =      g[i]*sqrt(t)   (dp/MonteCarloHost.c:180 -- the reference
 *                          dp CPU bug; kept ONLY so the unmodified dp object
 *                          pins stream / mat-vec / accumulation order bit for bit) */
void FN(orc_host_basket_paths)(int n, const REAL *s, const REAL *v, const REAL *p, const REAL *d,
                               const REAL *w, REAL k, REAL t, REAL r, int paths, unsigned seed,
                               int vol_in_diffusion, REAL *payoffs, orc_result *out)
{
    REAL sum = 0, sum2 = 0;
    REAL *g = (REAL *)malloc(sizeof(REAL) * (size_t)n * 2);
    REAL *bt = g + n;
    srand(seed);
    for (int i = 0; i < paths; i++) {
        for (int a = 0; a < n; a++)
            g[a] = FN(host_gaussian)(0, 1);
        for (int a = 0; a < n; a++) {
            REAL acc = 0;
            for (int b = 0; b < n; b++)
                acc += p[a * n + b] * g[b];
            bt[a] = acc;
        }
        for (int a = 0; a < n; a++)
            bt[a] += d[a];
        REAL basket = 0;
        for (int a = 0; a < n; a++) {
            REAL mu = (REAL)(((double)r - 0.5 * (double)v[a] * (double)v[a]) * (double)t);
            REAL si = vol_in_diffusion ? v[a] * bt[a] * SQRT_R(t) : bt[a] * SQRT_R(t);
            /* the reference first fills s[] then forms the weighted sum; same values */
            bt[a] = s[a] * EXP_R(mu + si);
        }
        for (int a = 0; a < n; a++)
            basket += bt[a] * w[a];
        REAL value = basket - k;
        REAL payoff = value > 0 ? value : 0;
        if (payoffs)
            payoffs[i] = payoff;
        sum += payoff;
        sum2 += payoff * payoff;
    }
    free(g);
    FN(host_close)(sum, sum2, paths, 1, r, t, out);
}
void FN(orc_host_basket)(int n, const REAL *s, const REAL *v, const REAL *p, const REAL *d,
                         const REAL *w, REAL k, REAL t, REAL r, int paths, unsigned seed,
                         int vol_in_diffusion, orc_result *out)
{
    FN(orc_host_basket_paths)(n, s, v, p, d, w, k, t, r, paths, seed, vol_in_diffusion, NULL, out);
}

/* CVA, HOST ordering: dp/MonteCarloHost.c:231-276.  Exposure at step j is priced at the
 * spot of step j-1 (the new spot is stored after the exposure, :254-261);
 * time to maturity by repeated subtraction, exposure zero once it goes negative (:255-259). */
void FN(orc_host_cva_paths)(REAL s0, REAL k, REAL r, REAL v, REAL t0, REAL defint, REAL lgd,
                            int n_grid, int paths, unsigned seed, REAL *values, orc_result *out)
{
    REAL sum = 0, sum2 = 0;
    REAL dt = t0 / n_grid;
    srand(seed);
    for (int i = 0; i < paths; i++) {
        REAL spot = s0, ttm = t0, acc = 0;
        for (int j = 1; j <= n_grid; j++) {
            REAL dpd = EXP_R(-(dt * (j - 1)) * defint) - EXP_R(-(dt * j) * defint);
            /* one GBM step over dt, dp/MonteCarloHost.c:164-167 */
            REAL g = FN(host_gaussian)(0, 1);
            REAL x = (REAL)(((double)r - 0.5 * (double)v * (double)v) * (double)dt +
                            (double)(g * SQRT_R(dt) * v));
            REAL next = spot * EXP_R(x);
            ttm -= dt;
            REAL ee = (ttm < 0) ? 0 : FN(orc_bs_call)(spot, k, r, v, ttm);
            acc += dpd * ee;
            spot = next;
        }
        acc *= lgd;
        if (values)
            values[i] = acc;
        sum += acc;
        sum2 += acc * acc;
    }
    FN(host_close)(sum, sum2, paths, 0, r, t0, out);
}
void FN(orc_host_cva)(REAL s0, REAL k, REAL r, REAL v, REAL t0, REAL defint, REAL lgd,
                      int n_grid, int paths, unsigned seed, orc_result *out)
{
    FN(orc_host_cva_paths)(s0, k, r, v, t0, defint, lgd, n_grid, paths, seed, NULL, out);
}

/* The reference's accumulation and closing applied to a given list of per-path values: sequential sums in REAL
 * (dp/MonteCarloHost.c:196-198,214-216,264-266), then :220-228 / :270-275.  With the values of orc_host_*_paths it
 * reproduces the reference's (Expected, Confidence) bit for bit; with the values of the orc_dev_*_on_normals functions
 * below it says what the reference would have printed had it evaluated the DEVICE formulas on its own normals. */
void FN(orc_ref_close)(const REAL *values, int paths, int discounted, REAL r, REAL t, orc_result *out)
{
    REAL sum = 0, sum2 = 0;
    for (int i = 0; i < paths; i++) {
        sum += values[i];
        sum2 += values[i] * values[i];
    }
    FN(host_close)(sum, sum2, paths, discounted, r, t, out);
}

/* ===================================================================================== */
/*  Product random stream (Philox4x32-10, counter-based) + reference DEVICE formulas       */
/* ===================================================================================== */

/* One block of the stream -> NPB normals (f32: one Philox block, 4; f64: three Philox blocks, 8) by two-branch Box-Muller.
 *   counter = { unit_hi, unit_lo, block, domain },  key = { seed_lo, seed_hi }   (mc_rng.hpp: philox_unit)
 * f32: radius uniform u_a = fma(x, 2^-32, 2^-33) in (0,1], angle u_b = 1 + (x >> 9) 2^-23 revolutions;  f64: see the body (8 normals from 3 Philox blocks)
 *   radius = sqrt(-2 ln u_a),  z_even = radius cos(2 pi u_b),  z_odd = radius sin(2 pi u_b)
 * The f32 radius is written with log2 (the HIP kernel's v_log_f32 is a base-2 log). */
static void FN(dev_normals_native)(uint64_t seed, uint32_t domain, uint64_t unit, uint32_t block, REAL *z)
{
    uint32_t x[4];
    /* Philox(counter, key), or the owning lane's XORWOW sequence; fp64 draws the block's other eight words below */
    orc_block_words(seed, domain, unit, ORC_IS_F32 ? block : 3 * block, x);
#if ORC_IS_F32
    for (int h = 0; h < 2; h++) {
        float ua = fmaf((float)x[2 * h], 0x1p-32f, 0x1p-33f);
        /* angle in revolutions: top 23 bits of the word as the mantissa of a float in [1, 2) (mc_rng.hpp: angle_f32) */
        uint32_t ub_bits = (x[2 * h + 1] >> 9) | 0x3f800000u;
        float ub;
        memcpy(&ub, &ub_bits, 4);
        float radius = sqrtf(-1.3862943611198906f * log2f(ua)); /* -2 ln 2 */
        double ang = 2.0 * M_PI * (double)ub;
        z[2 * h] = radius * (float)cos(ang);
        z[2 * h + 1] = radius * (float)sin(ang);
    }
#else
    /* fp64, stream version 2: block b = Philox blocks 3b, 3b + 1, 3b + 2 = twelve words W[0..11] = four Box-Muller pairs of
     * 96 bits (mc_rng.hpp: words_to_normals).  Pair p: a = W[3p], m = W[3p + 1], c = W[3p + 2];
     *   radius uniform  u_a = (J + 1/2) 2^-52,      J  = (a << 20) | (m >> 12)              52 bits
     *   angle uniform   u_b = (J' 2^8 + 1/2) 2^-52, J' = (c << 12) | (m & 0xfff)            44 bits */
    uint32_t W[12];
    memcpy(W, x, sizeof x);
    orc_block_words(seed, domain, unit, 3 * block + 1, W + 4);
    orc_block_words(seed, domain, unit, 3 * block + 2, W + 8);
    for (int p = 0; p < 4; p++) {
        uint32_t a = W[3 * p], m = W[3 * p + 1], c = W[3 * p + 2];
        uint64_t ka = ((uint64_t)a << 20) | (m >> 12);
        uint64_t kb = (((uint64_t)c << 12) | (m & 0xfffu)) << 8;
        double ua = ((double)ka + 0.5) * 0x1p-52;
        double ub = ((double)kb + 0.5) * 0x1p-52;
        double radius = sqrt(-2.0 * log(ua));
        double ang = 2.0 * M_PI * ub;
        z[2 * p] = radius * cos(ang);
        z[2 * p + 1] = radius * sin(ang);
    }
#endif
}

/* The normals of one block as the orc_dev_* family sees them, and how many there are:
 *   default               ORC_NPB native normals (above)
 *   orc_set_normals_f32   fp64 family only: FOUR fp32 normals of the block, widened -- the reference's own dp arithmetic,
 *                         `double z = curand_normal(...)` (dp/MonteCarloKernel.cu:68,78,250); twin of the
 *                         product's GenPhiloxF32N (mc_context_set_normals(ctx, MC_NORMALS_F32)). */
int FN(orc_dev_npb)(void)
{
#if !ORC_IS_F32
    if (orc_normals_f32_mode)
        return 4;
#endif
    return ORC_NPB;
}
void FN(orc_dev_normals)(uint64_t seed, uint32_t domain, uint64_t unit, uint32_t block, REAL *z)
{
#if !ORC_IS_F32
    if (orc_normals_f32_mode) {
        float f[4];
        dev_normals_native_f32(seed, domain, unit, block, f);
        for (int j = 0; j < 4; j++)
            z[j] = (double)f[j];
        return;
    }
#endif
    FN(dev_normals_native)(seed, domain, unit, block, z);
}

/* Normal source of the per-path formulas below: the product stream (block by block, each block fetched once and in
 * order -- what the XORWOW mode needs), or a caller-supplied array of `per_unit` normals per unit (the *_on_normals
 * entry points: the reference's own glibc stream pushed through the DEVICE formulas). */
typedef struct {
    const REAL *ext;
    uint64_t per_unit, first_unit;
    uint64_t seed;
    uint32_t domain;
    uint64_t have_unit;
    uint32_t have_block;
    int have;
    REAL z[8];
} FN(nsrc);

static FN(nsrc) FN(nsrc_stream)(uint64_t seed, uint32_t domain)
{
    FN(nsrc) q;
    memset(&q, 0, sizeof q);
    q.seed = seed, q.domain = domain;
    return q;
}
static FN(nsrc) FN(nsrc_external)(const REAL *z, uint64_t per_unit, uint64_t first_unit)
{
    FN(nsrc) q;
    memset(&q, 0, sizeof q);
    q.ext = z, q.per_unit = per_unit, q.first_unit = first_unit;
    return q;
}
/* normal number idx of unit `unit` */
static REAL FN(nsrc_get)(FN(nsrc) *q, uint64_t unit, uint32_t idx)
{
    if (q->ext)
        return idx < q->per_unit ? q->ext[(unit - q->first_unit) * q->per_unit + idx] : (REAL)0;
    const uint32_t npb = (uint32_t)FN(orc_dev_npb)(), block = idx / npb;
    if (!q->have || unit != q->have_unit || block != q->have_block) {
        FN(orc_dev_normals)(q->seed, q->domain, unit, block, q->z);
        q->have = 1, q->have_unit = unit, q->have_block = block;
    }
    return q->z[idx % npb];
}

static void FN(dev_finish)(double sum, double sum2, uint64_t n, double discount, orc_result *out)
{
    if (!out)
        return;
    out->sum = sum;
    out->sum2 = sum2;
    out->n = (long long)n;
    orc_closing(sum, sum2, (long long)n, discount, &out->expected, &out->confidence);
}

/* Vanilla, device formula dp/MonteCarloKernel.cu:67-71:
 *   payoff = max(S exp((r - v^2/2) T + v sqrt(T) z) - K, 0)
 * Path p draws normal (p mod NPB) of Philox unit (p div NPB), block 0, domain VANILLA.
 * Per-path values in REAL, (sum, sum2) accumulated in fp64. */
static void FN(dev_vanilla_core)(REAL s, REAL k, REAL r, REAL v, REAL t, FN(nsrc) *src, uint64_t npb,
                                 uint64_t first_path, uint64_t n_paths, int antithetic, REAL *payoffs, orc_result *out)
{
    const REAL drift = (REAL)(((double)r - 0.5 * (double)v * (double)v) * (double)t);
    const REAL vol = (REAL)((double)v * sqrt((double)t));
    double sum = 0, sum2 = 0;
    for (uint64_t i = 0; i < n_paths; i++) {
        uint64_t p = first_path + i;
        REAL z = FN(nsrc_get)(src, p / npb, (uint32_t)(p % npb));
        REAL value = s * EXP_R(drift + vol * z) - k;
        REAL payoff = value > 0 ? value : 0;
        if (antithetic) { /* sample = mean of the payoffs at z and -z */
            REAL mirror = s * EXP_R(drift - vol * z) - k;
            payoff = (REAL)0.5 * (payoff + (mirror > 0 ? mirror : 0));
        }
        if (payoffs)
            payoffs[i] = payoff;
        sum += (double)payoff;
        sum2 += (double)payoff * (double)payoff;
    }
    FN(dev_finish)(sum, sum2, n_paths, exp(-(double)r * (double)t), out);
}
void FN(orc_dev_vanilla)(REAL s, REAL k, REAL r, REAL v, REAL t, uint64_t seed,
                         uint64_t first_path, uint64_t n_paths, int antithetic, REAL *payoffs, orc_result *out)
{
    FN(nsrc) src = FN(nsrc_stream)(seed, ORC_DOMAIN_VANILLA);
    FN(dev_vanilla_core)(s, k, r, v, t, &src, (uint64_t)FN(orc_dev_npb)(), first_path, n_paths, antithetic, payoffs, out);
}
/* The same formula on a caller-supplied normal per path (z[i] prices path i): with the reference's own stream
 * (orc_host_gaussians) this bridges the device formulas to the compiled reference's outputs -- tests/test_oracle_bridge.py. */
void FN(orc_dev_vanilla_on_normals)(REAL s, REAL k, REAL r, REAL v, REAL t, const REAL *z, uint64_t n_paths,
                                    int antithetic, REAL *payoffs, orc_result *out)
{
    FN(nsrc) src = FN(nsrc_external)(z, 1, 0);
    FN(dev_vanilla_core)(s, k, r, v, t, &src, 1, 0, n_paths, antithetic, payoffs, out);
}

/* Basket, device formulas dp/MonteCarloKernel.cu:74-87 (bt = P g + d, P = Cholesky factor
 * held in the correlation slot) and :89-101 (s_j = S_j exp((r - v_j^2/2) T + v_j bt_j sqrt T),
 * payoff = max(sum_j w_j s_j - K, 0)).  Path p is Philox unit p; block b holds normals
 * g[b*NPB .. b*NPB+NPB-1]; domain BASKET.  The product multiplies only the lower triangle
 * (structural zeros of the factor skipped) -- identical values whenever the upper triangle is
 * zero, which Chol guarantees (dp/MonteCarloHost.c:95). */
/* Pathwise Greeks of the vanilla call on the product stream (not in the reference):
 * per path  S_T = S exp(drift + vol z),  I = [S_T > K]:  payoff I (S_T - K),  delta I S_T / S,
 * vega I S_T (sqrt(T) z - sigma T).  out[0..2] = price, delta, vega (each discounted). */
void FN(orc_dev_vanilla_greeks)(REAL s, REAL k, REAL r, REAL v, REAL t, uint64_t seed, uint64_t first_path,
                                uint64_t n_paths, orc_result *out)
{
    const REAL drift = (REAL)(((double)r - 0.5 * (double)v * (double)v) * (double)t);
    const REAL vol = (REAL)((double)v * sqrt((double)t));
    const REAL sqrt_t = (REAL)sqrt((double)t), sigma_t = (REAL)((double)v * (double)t);
    double acc[6] = {0, 0, 0, 0, 0, 0};
    REAL z[ORC_NPB];
    uint64_t have = (uint64_t)-1;
    for (uint64_t i = 0; i < n_paths; i++) {
        uint64_t p = first_path + i, unit = p / ORC_NPB;
        if (unit != have) {
            FN(dev_normals_native)(seed, ORC_DOMAIN_VANILLA, unit, 0, z);
            have = unit;
        }
        REAL zz = z[p % ORC_NPB];
        REAL st = s * EXP_R(drift + vol * zz);
        int itm = st > k;
        double pay = itm ? (double)(st - k) : 0.0, dl = itm ? (double)(st / s) : 0.0;
        double vg = itm ? (double)(st * (sqrt_t * zz - sigma_t)) : 0.0;
        acc[0] += pay, acc[1] += pay * pay, acc[2] += dl, acc[3] += dl * dl, acc[4] += vg, acc[5] += vg * vg;
    }
    for (int q = 0; q < 3; q++)
        FN(dev_finish)(acc[2 * q], acc[2 * q + 1], n_paths, exp(-(double)r * (double)t), out + q);
}

/* Likelihood-ratio Greeks of the vanilla call (not in the reference): the score of the lognormal density
 * times the payoff.  delta = payoff z / (S sigma sqrt T),  vega = payoff ((z^2 - 1) / sigma - z sqrt T).
 * out[0..2] = price, delta, vega (each discounted). */
void FN(orc_dev_vanilla_greeks_lr)(REAL s, REAL k, REAL r, REAL v, REAL t, uint64_t seed, uint64_t first_path,
                                   uint64_t n_paths, orc_result *out)
{
    const REAL drift = (REAL)(((double)r - 0.5 * (double)v * (double)v) * (double)t);
    const REAL vol = (REAL)((double)v * sqrt((double)t));
    const REAL sqrt_t = (REAL)sqrt((double)t);
    const REAL lr_delta = (REAL)(1.0 / ((double)s * (double)v * sqrt((double)t))), inv_sigma = (REAL)(1.0 / (double)v);
    double acc[6] = {0, 0, 0, 0, 0, 0};
    REAL z[ORC_NPB];
    uint64_t have = (uint64_t)-1;
    for (uint64_t i = 0; i < n_paths; i++) {
        uint64_t p = first_path + i, unit = p / ORC_NPB;
        if (unit != have) {
            FN(dev_normals_native)(seed, ORC_DOMAIN_VANILLA, unit, 0, z);
            have = unit;
        }
        REAL zz = z[p % ORC_NPB];
        REAL st = s * EXP_R(drift + vol * zz);
        REAL payoff = st > k ? st - k : 0;
        double pay = (double)payoff, dl = (double)(payoff * zz * lr_delta);
        double vg = (double)(payoff * ((zz * zz - (REAL)1) * inv_sigma - zz * sqrt_t));
        acc[0] += pay, acc[1] += pay * pay, acc[2] += dl, acc[3] += dl * dl, acc[4] += vg, acc[5] += vg * vg;
    }
    for (int q = 0; q < 3; q++)
        FN(dev_finish)(acc[2 * q], acc[2 * q + 1], n_paths, exp(-(double)r * (double)t), out + q);
}

/* Pathwise Greeks of the basket call, reference device formulas dp/MonteCarloKernel.cu:74-101:
 * B = sum_a w_a s_a, I = [B > K]; delta_a = I w_a s_a / S_a, vega_a = I w_a s_a (bt_a sqrt T - v_a T).
 * out[0] = price, out[1 + a] = delta_a, out[1 + n + a] = vega_a (each discounted). */
void FN(orc_dev_basket_greeks)(int n, const REAL *s, const REAL *v, const REAL *p, const REAL *d, const REAL *w, REAL k,
                               REAL t, REAL r, uint64_t seed, uint64_t first_path, uint64_t n_paths, orc_result *out)
{
    int nblk = (n + ORC_NPB - 1) / ORC_NPB;
    REAL *g = (REAL *)malloc(sizeof(REAL) * (size_t)(nblk * ORC_NPB));
    REAL *term = (REAL *)malloc(sizeof(REAL) * (size_t)n), *bts = (REAL *)malloc(sizeof(REAL) * (size_t)n);
    double *acc = (double *)calloc((size_t)(2 * (1 + 2 * n)), sizeof(double));
    const REAL sqrt_t = (REAL)sqrt((double)t);
    for (uint64_t i = 0; i < n_paths; i++) {
        uint64_t path = first_path + i;
        for (int b = 0; b < nblk; b++)
            FN(dev_normals_native)(seed, ORC_DOMAIN_BASKET, path, (uint32_t)b, g + b * ORC_NPB);
        REAL basket = 0;
        for (int a = 0; a < n; a++) {
            REAL bt = 0;
            for (int b = 0; b <= a; b++)
                bt += p[a * n + b] * g[b];
            bt += d[a];
            REAL mu = (REAL)(((double)r - 0.5 * (double)v[a] * (double)v[a]) * (double)t);
            term[a] = s[a] * EXP_R(mu + v[a] * bt * sqrt_t) * w[a];
            bts[a] = bt;
            basket += term[a];
        }
        int itm = basket > k;
        double pay = itm ? (double)(basket - k) : 0.0;
        acc[0] += pay, acc[1] += pay * pay;
        for (int a = 0; a < n; a++) {
            double dl = itm ? (double)(term[a] * (REAL)(1.0 / (double)s[a])) : 0.0;
            double vg = itm ? (double)(term[a] * (bts[a] * sqrt_t - (REAL)((double)v[a] * (double)t))) : 0.0;
            acc[2 * (1 + a)] += dl, acc[2 * (1 + a) + 1] += dl * dl;
            acc[2 * (1 + n + a)] += vg, acc[2 * (1 + n + a) + 1] += vg * vg;
        }
    }
    for (int q = 0; q < 1 + 2 * n; q++)
        FN(dev_finish)(acc[2 * q], acc[2 * q + 1], n_paths, exp(-(double)r * (double)t), out + q);
    free(g), free(term), free(bts), free(acc);
}

/* Likelihood-ratio Greeks of the basket call: twin of basket_greeks_kernel<Real, true> (csrc/mc_kernels.hpp).  The payoff
 * times the score of the terminal prices' joint lognormal density; y = L^-T g with M = L^-T formed in fp64 by back-substitution and
 * rounded once, as the product's host side does (csrc/mc_api.hip: basket_greeks_run):
 *   delta_a = payoff y_a / (S_a v_a sqrt T),   vega_a = payoff [ (y_a (L g)_a - 1) / v_a + (sqrt T d_a - v_a T) y_a / (v_a sqrt T) ]
 * out as in orc_dev_basket_greeks.  Returns 0, or -1 when the factor is singular (some p[a][a] <= 0). */
int FN(orc_dev_basket_greeks_lr)(int n, const REAL *s, const REAL *v, const REAL *p, const REAL *d, const REAL *w, REAL k,
                                 REAL t, REAL r, uint64_t seed, uint64_t first_path, uint64_t n_paths, orc_result *out)
{
    int nblk = (n + ORC_NPB - 1) / ORC_NPB;
    const double sqrt_td = sqrt((double)t);
    double *inv = (double *)calloc((size_t)n * (size_t)n, sizeof(double));
    for (int a = 0; a < n; a++) {
        double laa = (double)p[a * n + a];
        if (!(laa > 0)) {
            free(inv);
            return -1;
        }
        for (int b = 0; b <= a; b++) {
            double acc = a == b ? 1.0 : 0.0;
            for (int q = b; q < a; q++)
                acc -= (double)p[a * n + q] * inv[q * n + b];
            inv[a * n + b] = acc / laa;
        }
    }
    REAL *M = (REAL *)calloc((size_t)n * (size_t)n, sizeof(REAL));
    REAL *inv_svt = (REAL *)malloc(sizeof(REAL) * (size_t)n), *inv_v = (REAL *)malloc(sizeof(REAL) * (size_t)n), *mcoef = (REAL *)malloc(sizeof(REAL) * (size_t)n);
    for (int a = 0; a < n; a++) {
        for (int b = a; b < n; b++)
            M[a * n + b] = (REAL)inv[b * n + a];
        inv_svt[a] = (REAL)(1.0 / ((double)s[a] * (double)v[a] * sqrt_td));
        inv_v[a] = (REAL)(1.0 / (double)v[a]);
        mcoef[a] = (REAL)((sqrt_td * (double)d[a] - (double)v[a] * (double)t) / ((double)v[a] * sqrt_td));
    }
    REAL *g = (REAL *)malloc(sizeof(REAL) * (size_t)(nblk * ORC_NPB));
    REAL *bt0 = (REAL *)malloc(sizeof(REAL) * (size_t)n);
    double *acc = (double *)calloc((size_t)(2 * (1 + 2 * n)), sizeof(double));
    const REAL sqrt_t = (REAL)sqrt_td;
    for (uint64_t i = 0; i < n_paths; i++) {
        uint64_t path = first_path + i;
        for (int b = 0; b < nblk; b++)
            FN(dev_normals_native)(seed, ORC_DOMAIN_BASKET, path, (uint32_t)b, g + b * ORC_NPB);
        REAL basket = 0;
        for (int a = 0; a < n; a++) {
            REAL bt = 0;
            for (int b = 0; b <= a; b++)
                bt += p[a * n + b] * g[b];
            bt0[a] = bt;
            bt += d[a];
            REAL mu = (REAL)(((double)r - 0.5 * (double)v[a] * (double)v[a]) * (double)t);
            basket += s[a] * EXP_R(mu + v[a] * bt * sqrt_t) * w[a];
        }
        REAL payoff = basket > k ? basket - k : 0;
        double pay = (double)payoff;
        acc[0] += pay, acc[1] += pay * pay;
        for (int a = 0; a < n; a++) {
            REAL y = 0;
            for (int b = a; b < n; b++)
                y += M[a * n + b] * g[b];
            double dl = (double)(payoff * (y * inv_svt[a]));
            double vg = (double)(payoff * ((y * bt0[a] - 1) * inv_v[a] + mcoef[a] * y));
            acc[2 * (1 + a)] += dl, acc[2 * (1 + a) + 1] += dl * dl;
            acc[2 * (1 + n + a)] += vg, acc[2 * (1 + n + a) + 1] += vg * vg;
        }
    }
    for (int q = 0; q < 1 + 2 * n; q++)
        FN(dev_finish)(acc[2 * q], acc[2 * q + 1], n_paths, exp(-(double)r * (double)t), out + q);
    free(inv), free(M), free(inv_svt), free(inv_v), free(mcoef), free(g), free(bt0), free(acc);
    return 0;
}

/* Closed-form mean of the geometric-basket control (not in the reference):
 *   G = W prod_a S_a(T)^(w_a / W),  W = sum_a w_a > 0,  is lognormal: ln G ~ N(m, sd^2) with
 *   m = ln W + sum_a wh_a (ln S_a + (r - v_a^2/2) T + v_a sqrt(T) d_a),  wh = w / W,
 *   sd^2 = sum_b (sum_{a>=b} wh_a v_a sqrt(T) L_ab)^2,
 *   E[max(G - K, 0)] = e^{m + sd^2/2} Phi(d1) - K Phi(d2),  d1 = (m - ln K + sd^2)/sd, d2 = d1 - sd.
 * Evaluated in fp64 whatever the simulation precision. */
double FN(orc_basket_control_mean)(int n, const REAL *s, const REAL *v, const REAL *p, const REAL *d,
                                   const REAL *w, REAL k, REAL t, REAL r)
{
    double W = 0;
    for (int a = 0; a < n; a++)
        W += (double)w[a];
    const double sqrt_t = sqrt((double)t);
    double m = log(W), var = 0;
    for (int a = 0; a < n; a++)
        m += (double)w[a] / W * (log((double)s[a]) + ((double)r - 0.5 * (double)v[a] * (double)v[a]) * (double)t +
                                 (double)v[a] * sqrt_t * (double)d[a]);
    for (int b = 0; b < n; b++) {
        double c = 0;
        for (int a = b; a < n; a++)
            c += (double)w[a] / W * (double)v[a] * sqrt_t * (double)p[a * n + b];
        var += c * c;
    }
    const double sd = sqrt(var);
    if (sd == 0) {
        double g = exp(m) - (double)k;
        return g > 0 ? g : 0;
    }
    const double d1 = (m - log((double)k) + var) / sd, d2 = d1 - sd;
    return exp(m + 0.5 * var) * 0.5 * erfc(-d1 / sqrt(2.0)) - (double)k * 0.5 * erfc(-d2 / sqrt(2.0));
}

/* `mode` bit 0: antithetic variates; bit 1: geometric-basket control variate (the per-path value is
 * then payoff(arithmetic) - payoff(geometric), and the closed-form mean above is added back to the
 * expectation); bit 2 (bridge tests only): the diffusion WITHOUT the volatility, si = bt_a sqrt T -- what the
 * reference's dp CPU path computes (dp/MonteCarloHost.c:180), so that the unmodified dp object can pin
 * the device formulas' mat-vec and accumulation order bit for bit. */
static void FN(dev_basket_core)(int n, const REAL *s, const REAL *v, const REAL *p, const REAL *d,
                                const REAL *w, REAL k, REAL t, REAL r, FN(nsrc) *src, int n_draw,
                                uint64_t first_path, uint64_t n_paths, int mode, REAL *payoffs, orc_result *out)
{
    const int antithetic = mode & 1, control = (mode >> 1) & 1, no_vol = (mode >> 2) & 1;
    REAL *g = (REAL *)malloc(sizeof(REAL) * (size_t)(n_draw > n ? n_draw : n));
    const REAL sqrt_t = (REAL)sqrt((double)t);
    double wsum = 0;
    for (int a = 0; a < n; a++)
        wsum += (double)w[a];
    double sum = 0, sum2 = 0;
    for (uint64_t i = 0; i < n_paths; i++) {
        uint64_t path = first_path + i;
        /* n_draw >= n normals are drawn (whole blocks; in XORWOW mode the generic kernel's padding to a multiple of 4
         * too: the lane's sequence moves on) */
        for (int b = 0; b < n_draw; b++)
            g[b] = FN(nsrc_get)(src, path, (uint32_t)b);
        REAL payoff = 0;
        for (int sign = 1; sign >= (antithetic ? -1 : 1); sign -= 2) {
            REAL basket = 0, lg = (REAL)log(wsum);
            for (int a = 0; a < n; a++) {
                REAL bt = 0;
                for (int b = 0; b <= a; b++)
                    bt += p[a * n + b] * ((REAL)sign * g[b]);
                bt += d[a];
                REAL mu = (REAL)(((double)r - 0.5 * (double)v[a] * (double)v[a]) * (double)t);
                REAL si = no_vol ? bt * sqrt_t : v[a] * bt * sqrt_t;
                REAL sa = s[a] * EXP_R(mu + si);
                basket += sa * w[a];
                lg += (REAL)((double)w[a] / wsum) * (LOG_R(s[a]) + (mu + si));
            }
            REAL value = basket - k;
            payoff += value > 0 ? value : 0;
            if (control) {
                REAL gv = EXP_R(lg) - k;
                payoff -= gv > 0 ? gv : 0;
            }
        }
        if (antithetic)
            payoff *= (REAL)0.5;
        if (payoffs)
            payoffs[i] = payoff;
        sum += (double)payoff;
        sum2 += (double)payoff * (double)payoff;
    }
    free(g);
    const double disc = exp(-(double)r * (double)t);
    FN(dev_finish)(sum, sum2, n_paths, disc, out);
    if (control && out)
        out->expected += disc * FN(orc_basket_control_mean)(n, s, v, p, d, w, k, t, r);
}
void FN(orc_dev_basket)(int n, const REAL *s, const REAL *v, const REAL *p, const REAL *d,
                        const REAL *w, REAL k, REAL t, REAL r, uint64_t seed,
                        uint64_t first_path, uint64_t n_paths, int mode, REAL *payoffs, orc_result *out)
{
    const int npb = FN(orc_dev_npb)();
    int n_draw = (n + npb - 1) / npb * npb;
    if (orc_xorwow_active())   /* the generic kernel pads n to a multiple of 4, then to whole blocks: the lane's sequence moves on */
        n_draw = (4 * ((n + 3) / 4) + npb - 1) / npb * npb;
    FN(nsrc) src = FN(nsrc_stream)(seed, ORC_DOMAIN_BASKET);
    FN(dev_basket_core)(n, s, v, p, d, w, k, t, r, &src, n_draw, first_path, n_paths, mode & 3, payoffs, out);
}
/* The same formulas on a caller-supplied stream, n normals per path in drawing order (g[i * n + a]: path i, asset a) --
 * the order the reference's simGaussVect draws them (dp/MonteCarloHost.c:150-161). */
void FN(orc_dev_basket_on_normals)(int n, const REAL *s, const REAL *v, const REAL *p, const REAL *d,
                                   const REAL *w, REAL k, REAL t, REAL r, const REAL *g, uint64_t n_paths,
                                   int mode, REAL *payoffs, orc_result *out)
{
    FN(nsrc) src = FN(nsrc_external)(g, (uint64_t)n, 0);
    FN(dev_basket_core)(n, s, v, p, d, w, k, t, r, &src, n, 0, n_paths, mode, payoffs, out);
}

/* CVA, DEVICE ordering dp/MonteCarloKernel.cu:241-262: at step j the spot is advanced FIRST
 * and the exposure is the Black-Scholes value at the NEW spot and the new time to maturity
 * (SURVEY 2.3 #7).  Time to maturity by repeated subtraction in REAL; once negative the
 * exposure is 0 and no normal is consumed (:249-256).
 * Product semantics stated in DESIGN.md and mirrored here:
 *   - default-probability increment dp_j = e^{-lambda t_{j-1}} - e^{-lambda t_j},
 *     t_j = dt*j, evaluated in fp64 via expm1 and rounded to REAL;
 *   - residual maturity exactly 0 -> exposure = intrinsic max(s-K,0), the limit of the
 *     closed form, instead of 0/0;
 *   - step j uses normal (j-1): Philox unit = path, block = (j-1) div NPB, domain CVA;
 *   - result is LGD * sum_j dp_j ee_j, NOT discounted (dp/MonteCarloKernel.cu:259,466).
 * `flags` (bridge tests only; 0 = the product's semantics above):
 *   ORC_CVA_HOST_ORDER  the reference CPU loop's ordering (dp/MonteCarloHost.c:252-262): a normal is drawn at EVERY date,
 *                       and the exposure of date j is priced at the spot of date j-1
 *   ORC_CVA_REF_DP      dp_j as the reference forms it, a difference of two exponentials in REAL (:253, device :248)
 *   ORC_CVA_REF_T0      a date with residual maturity exactly 0 goes through the closed form like any other (0/0 and all) */
static void FN(dev_cva_core)(REAL s0, REAL k, REAL r, REAL v, REAL t0, REAL defint, REAL lgd, int n_grid,
                             FN(nsrc) *src, uint64_t first_path, uint64_t n_paths, int antithetic, int flags,
                             REAL *values, orc_result *out)
{
    const int host_order = flags & ORC_CVA_HOST_ORDER, ref_dp = flags & ORC_CVA_REF_DP, ref_t0 = flags & ORC_CVA_REF_T0;
    const REAL dt = t0 / n_grid;
    const REAL step_drift = (REAL)(((double)r - 0.5 * (double)v * (double)v) * (double)dt);
    const REAL step_vol = (REAL)((double)v * sqrt((double)dt));
    double sum = 0, sum2 = 0;
    for (uint64_t i = 0; i < n_paths; i++) {
        uint64_t path = first_path + i;
        REAL spot = s0, mirror = s0, ttm = t0, acc = 0;
        for (int j = 1; j <= n_grid; j++) {
            double t_prev = (double)dt * (double)(j - 1), t_now = (double)dt * (double)j;
            /* e^{-l a} - e^{-l b} = -e^{-l a} expm1(-l (b - a)) */
            REAL dpd = ref_dp ? EXP_R(-(dt * (j - 1)) * defint) - EXP_R(-(dt * j) * defint)
                              : (REAL)(-exp(-(double)defint * t_prev) * expm1(-(double)defint * (t_now - t_prev)));
            REAL ee = 0;
            REAL spot_was = spot, mirror_was = mirror;
            ttm -= dt;
            if (ttm >= 0 || host_order) {
                REAL z = FN(nsrc_get)(src, path, (uint32_t)(j - 1));
                spot = spot * EXP_R(step_drift + step_vol * z);
                mirror = mirror * EXP_R(step_drift - step_vol * z);
            }
            if (ttm >= 0) {
                for (int leg = 0; leg < (antithetic ? 2 : 1); leg++) {
                    REAL sx = host_order ? (leg ? mirror_was : spot_was) : (leg ? mirror : spot), e1;
                    if (ttm == 0 && !ref_t0) {
                        REAL iv = sx - k;
                        e1 = iv > 0 ? iv : 0;
                    } else {
                        e1 = FN(orc_bs_call)(sx, k, r, v, ttm);
                    }
                    ee += e1;
                }
                if (antithetic)
                    ee *= (REAL)0.5;
            }
            acc += dpd * ee;
        }
        acc *= lgd;
        if (values)
            values[i] = acc;
        sum += (double)acc;
        sum2 += (double)acc * (double)acc;
    }
    FN(dev_finish)(sum, sum2, n_paths, 1.0, out);
}
void FN(orc_dev_cva)(REAL s0, REAL k, REAL r, REAL v, REAL t0, REAL defint, REAL lgd, int n_grid,
                     uint64_t seed, uint64_t first_path, uint64_t n_paths, int antithetic, REAL *values,
                     orc_result *out)
{
    FN(nsrc) src = FN(nsrc_stream)(seed, ORC_DOMAIN_CVA);
    FN(dev_cva_core)(s0, k, r, v, t0, defint, lgd, n_grid, &src, first_path, n_paths, antithetic, 0, values, out);
}
/* The same loop on a caller-supplied stream, n_grid normals per path (z[i * n_grid + j - 1]: path i, date j). */
void FN(orc_dev_cva_on_normals)(REAL s0, REAL k, REAL r, REAL v, REAL t0, REAL defint, REAL lgd, int n_grid,
                                const REAL *z, uint64_t n_paths, int antithetic, int flags, REAL *values, orc_result *out)
{
    FN(nsrc) src = FN(nsrc_external)(z, (uint64_t)n_grid, 0);
    FN(dev_cva_core)(s0, k, r, v, t0, defint, lgd, n_grid, &src, 0, n_paths, antithetic, flags, values, out);
}

/* CVA with its pathwise delta and vega: the loop of orc_dev_cva (plain estimator) carrying
 *   d CVA / d S_0   = LGD sum_j dp_j Delta_j S_j / S_0,   Delta_j = cnd(d1_j) of the reference's closed form
 *                     (dp/MonteCarloKernel.cu:110-129), I[S_j > K] on a date with residual maturity exactly 0;
 *   d CVA / d sigma = LGD sum_j dp_j [ S_j phi(d1_j) sqrt(tau_j) + Delta_j S_j (W_j sqrt(dt) - sigma t_j) ],
 *                     W_j = z_1 + ... + z_j (the closed form's vega + the path's own sensitivity to sigma).
 * out[0] = CVA, out[1] = delta, out[2] = vega. */
void FN(orc_dev_cva_greeks)(REAL s0, REAL k, REAL r, REAL v, REAL t0, REAL defint, REAL lgd, int n_grid, uint64_t seed,
                            uint64_t first_path, uint64_t n_paths, orc_result *out)
{
    const REAL dt = t0 / n_grid;
    const REAL step_drift = (REAL)(((double)r - 0.5 * (double)v * (double)v) * (double)dt);
    const REAL step_vol = (REAL)((double)v * sqrt((double)dt));
    const REAL sqrt_dt = (REAL)sqrt((double)dt);
    double acc[6] = {0, 0, 0, 0, 0, 0};
    REAL z[ORC_NPB];
    for (uint64_t i = 0; i < n_paths; i++) {
        uint64_t path = first_path + i;
        REAL spot = s0, ttm = t0, cva = 0, delta = 0, vega = 0, wsum = 0;
        for (int j = 1; j <= n_grid; j++) {
            double t_prev = (double)dt * (double)(j - 1), t_now = (double)dt * (double)j;
            REAL dpd = (REAL)(-exp(-(double)defint * t_prev) * expm1(-(double)defint * (t_now - t_prev)));
            ttm -= dt;
            if (!(ttm >= 0))
                continue;
            int idx = j - 1;
            if (idx % ORC_NPB == 0)
                FN(dev_normals_native)(seed, ORC_DOMAIN_CVA, path, (uint32_t)(idx / ORC_NPB), z);
            spot = spot * EXP_R(step_drift + step_vol * z[idx % ORC_NPB]);
            wsum += z[idx % ORC_NPB];
            REAL ee, sd, sphi = 0;
            if (ttm == 0) {
                ee = spot > k ? spot - k : 0;
                sd = spot > k ? spot : 0;
            } else {
                REAL sqrt_t = SQRT_R(ttm);
                double num = (double)LOG_R(spot / k) + ((double)r + 0.5 * (double)v * (double)v) * (double)ttm;
                REAL d1 = (REAL)(num / (double)(v * sqrt_t));
                ee = FN(orc_bs_call)(spot, k, r, v, ttm);
                sd = spot * FN(orc_cnd)(d1);
                sphi = spot * (REAL)0.39894228040143267793994605993438 * EXP_R((REAL)(-0.5 * (double)d1 * (double)d1));
            }
            cva += dpd * ee;
            delta += dpd * sd;
            vega += dpd * (sphi * SQRT_R(ttm) + sd * (wsum * sqrt_dt - (REAL)((double)v * t_now)));
        }
        double c = (double)(cva * lgd), dl = (double)(delta * lgd * (REAL)(1.0 / (double)s0)), vg = (double)(vega * lgd);
        acc[0] += c, acc[1] += c * c, acc[2] += dl, acc[3] += dl * dl, acc[4] += vg, acc[5] += vg * vg;
    }
    FN(dev_finish)(acc[0], acc[1], n_paths, 1.0, out);
    FN(dev_finish)(acc[2], acc[3], n_paths, 1.0, out + 1);
    FN(dev_finish)(acc[4], acc[5], n_paths, 1.0, out + 2);
}

/* Likelihood-ratio Greeks of the CVA: twin of cva_greeks_kernel<Real, true>.  Only the first transition's density depends on S_0,
 * every transition's on sigma; the closed-form exposure's own dependence on sigma stays pathwise (S_j phi(d1_j) sqrt(tau_j)):
 *   delta = CVA_path z_1 / (S_0 sigma sqrt dt),   vega = LGD sum_j dp_j S_j phi(d1_j) sqrt(tau_j) + CVA_path sum_j ((z_j^2 - 1) / sigma - z_j sqrt dt)
 * over the dates that draw a normal.  out[0] = CVA, out[1] = delta, out[2] = vega. */
void FN(orc_dev_cva_greeks_lr)(REAL s0, REAL k, REAL r, REAL v, REAL t0, REAL defint, REAL lgd, int n_grid, uint64_t seed,
                               uint64_t first_path, uint64_t n_paths, orc_result *out)
{
    const REAL dt = t0 / n_grid;
    const REAL step_drift = (REAL)(((double)r - 0.5 * (double)v * (double)v) * (double)dt);
    const REAL step_vol = (REAL)((double)v * sqrt((double)dt));
    const REAL sqrt_dt = (REAL)sqrt((double)dt);
    const REAL lr_delta = (REAL)(1.0 / ((double)s0 * (double)v * sqrt((double)dt))), inv_sigma = (REAL)(1.0 / (double)v);
    double acc[6] = {0, 0, 0, 0, 0, 0};
    REAL z[ORC_NPB];
    for (uint64_t i = 0; i < n_paths; i++) {
        uint64_t path = first_path + i;
        REAL spot = s0, ttm = t0, cva = 0, vega = 0, score = 0, z_first = 0;
        for (int j = 1; j <= n_grid; j++) {
            double t_prev = (double)dt * (double)(j - 1), t_now = (double)dt * (double)j;
            REAL dpd = (REAL)(-exp(-(double)defint * t_prev) * expm1(-(double)defint * (t_now - t_prev)));
            ttm -= dt;
            if (!(ttm >= 0))
                continue;
            int idx = j - 1;
            if (idx % ORC_NPB == 0)
                FN(dev_normals_native)(seed, ORC_DOMAIN_CVA, path, (uint32_t)(idx / ORC_NPB), z);
            REAL zz = z[idx % ORC_NPB];
            spot = spot * EXP_R(step_drift + step_vol * zz);
            if (idx == 0)
                z_first = zz;
            score += (zz * zz - 1) * inv_sigma - zz * sqrt_dt;
            REAL ee, sphi = 0;
            if (ttm == 0) {
                ee = spot > k ? spot - k : 0;
            } else {
                REAL sqrt_t = SQRT_R(ttm);
                double num = (double)LOG_R(spot / k) + ((double)r + 0.5 * (double)v * (double)v) * (double)ttm;
                REAL d1 = (REAL)(num / (double)(v * sqrt_t));
                ee = FN(orc_bs_call)(spot, k, r, v, ttm);
                sphi = spot * (REAL)0.39894228040143267793994605993438 * EXP_R((REAL)(-0.5 * (double)d1 * (double)d1));
            }
            cva += dpd * ee;
            vega += dpd * (sphi * SQRT_R(ttm));
        }
        REAL cva_path = cva * lgd;
        double c = (double)cva_path, dl = (double)(cva_path * (z_first * lr_delta)), vg = (double)(cva_path * score + vega * lgd);
        acc[0] += c, acc[1] += c * c, acc[2] += dl, acc[3] += dl * dl, acc[4] += vg, acc[5] += vg * vg;
    }
    FN(dev_finish)(acc[0], acc[1], n_paths, 1.0, out);
    FN(dev_finish)(acc[2], acc[3], n_paths, 1.0, out + 1);
    FN(dev_finish)(acc[4], acc[5], n_paths, 1.0, out + 2);
}

#undef FN
#undef ORC_CAT
#undef ORC_CAT_
