/* mc_hostmath_impl.h -- precision-generic body of mc_hostmath.c (REAL, SQRT_R, X, BASKET set by the includer). */

/* Cholesky with the reference's semantics (dp/MonteCarloHost.c:90-105: column by column, a non-positive pivot leaves
 * its column zero).  Returns the number of non-positive pivots met. */
int FN(mc_chol)(int n, const REAL *c, REAL *a)
{
    if (n < 1 || !c || !a)
        return -1;
    REAL *work = (REAL *)malloc(sizeof(REAL) * (size_t)n);
    if (!work)
        return -1;
    int bad = 0;
    for (int col = 0; col < n; ++col) {
        for (int row = 0; row < n; ++row) {
            a[row * n + col] = 0;
            if (row < col)
                continue;
            work[row] = c[row * n + col];
            for (int q = 0; q < col; ++q)
                work[row] -= a[col * n + q] * a[row * n + q];
            if (work[col] > 0)
                a[row * n + col] = work[row] / SQRT_R(work[col]);
            else if (row == col)
                ++bad;
        }
    }
    free(work);
    return bad;
}

/* Covariance input.  The reference's drivers hold volatilities and a CORRELATION matrix and factor
 * the latter with Chol before either path runs (dp/basketOpt.cu:34-61,96-99); a caller who holds the covariance of
 * the annualised log-returns gets both inputs of the basket structs from it here: v_a = sqrt(cov_aa), correlation
 * cov_ab / (v_a v_b) with an exact unit diagonal, then Chol's factorisation (same arithmetic, same zero-pivot rule,
 * dp/MonteCarloHost.c:90-105).  Like Chol, only the lower triangle of the input is read.  All in REAL. */
int FN(mc_factor_from_cov)(int n, const REAL *cov, REAL *v, REAL *p)
{
    if (n < 1 || !cov || !v || !p)
        return -1;
    for (int a = 0; a < n; ++a) {
        const REAL var = cov[a * n + a];
        if (!(var > 0) || !isfinite((double)var))
            return -1;   /* no volatility to extract */
        v[a] = SQRT_R(var);
    }
    REAL *corr = (REAL *)calloc((size_t)n * (size_t)n, sizeof(REAL));
    if (!corr)
        return -1;
    for (int a = 0; a < n; ++a) {
        corr[(size_t)a * n + a] = 1;
        for (int b = 0; b < a; ++b) {
            const REAL c = cov[a * n + b];
            if (!isfinite((double)c)) {
                free(corr);
                return -1;
            }
            corr[(size_t)a * n + b] = corr[(size_t)b * n + a] = c / (v[a] * v[b]);
        }
    }
    const int bad = FN(mc_chol)(n, corr, p);
    free(corr);
    return bad;
}

/* E[max(G - K, 0)] of the geometric-basket control, closed form in fp64 (see mc_mi355x.h) */
int FN(mc_basket_control_mean)(const BASKET *o, double *mean)
{
    if (!o)
        return mc_internal_fail(MC_ERR_INVALID, "NULL basket");
    if (!mean || o->n < 1 || o->n > MC_MAX_ASSETS_GENERIC || !o->s || !o->v || !o->p || !o->d || !o->w)
        return mc_internal_fail(MC_ERR_INVALID, "control variate: bad basket");
    const int n = o->n;
    double W = 0;
    for (int a = 0; a < n; ++a) {
        if (!((double)o->w[a] > 0) || !((double)o->s[a] > 0))
            return mc_internal_fail(MC_ERR_INVALID, "control variate: needs w[a] > 0 and s[a] > 0 for every asset");
        W += (double)o->w[a];
    }
    if (!((double)o->k > 0))
        return mc_internal_fail(MC_ERR_INVALID, "control variate: needs k > 0");
    const double sqrt_t = sqrt((double)o->t);
    double m = log(W), var = 0;
    for (int a = 0; a < n; ++a) {
        const double va = (double)o->v[a];
        m += (double)o->w[a] / W * (log((double)o->s[a]) + ((double)o->r - 0.5 * va * va) * (double)o->t + va * sqrt_t * (double)o->d[a]);
    }
    for (int b = 0; b < n; ++b) {
        double cb = 0;
        for (int a = b; a < n; ++a)
            cb += (double)o->w[a] / W * (double)o->v[a] * sqrt_t * (double)o->p[a * n + b];
        var += cb * cb;
    }
    const double sd = sqrt(var);
    if (sd == 0) {
        const double g = exp(m) - (double)o->k;
        *mean = g > 0 ? g : 0;
        return MC_OK;
    }
    const double d1 = (m - log((double)o->k) + var) / sd, d2 = d1 - sd;
    *mean = exp(m + 0.5 * var) * 0.5 * erfc(-d1 / sqrt(2.0)) - (double)o->k * 0.5 * erfc(-d2 / sqrt(2.0));
    return MC_OK;
}
