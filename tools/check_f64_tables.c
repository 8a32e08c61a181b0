/* Host check of the table-driven fp64 Box-Muller pieces (same arithmetic as mc_math_f64.hpp, written with
 * C99 fma): max error against 80-bit long double libm over random and edge inputs.
 *   gcc -O2 -ffp-contract=off -o /tmp/check_f64_tables tools/check_f64_tables.c -lm && /tmp/check_f64_tables */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double x, y; } d2;
static const d2 TAB[512] = {
#include "../montecarlocuda_amd/csrc/mc_tables_f64.inc"
};

static double mk(uint32_t hi, uint32_t lo) { uint64_t b = ((uint64_t)hi << 32) | lo; double d; memcpy(&d, &b, 8); return d; }
static void split(double d, uint32_t *hi, uint32_t *lo) { uint64_t b; memcpy(&b, &d, 8); *hi = b >> 32; *lo = (uint32_t)b; }

static double u01(uint32_t lo, uint32_t hi)
{
    return mk((hi >> 12) | 0x3ff00000u, (hi << 20) | (lo >> 12)) + (-1.0 + 0x1p-53);
}

static double neg2log_unit_tab(double u)
{
    uint32_t hi, lo;
    split(u, &hi, &lo);
    const int32_t h = (int32_t)(hi - 0x3fe6a09eu);
    const int32_t k = h >> 20;
    const double m = mk(hi - ((uint32_t)k << 20), lo);
    const d2 e = TAB[(h >> 13) & 0x7f];
    const double r = fma(m, e.x, 2.0);
    double p = fma(r, 1.0 / 448, 1.0 / 192);
    p = fma(r, p, 1.0 / 80);
    p = fma(r, p, 1.0 / 32);
    p = fma(r, p, 1.0 / 12);
    p = fma(r, p, 0.25);
    const double small = fma(r * r, p, r);
    const double big = fma((double)k, -1.3862943611198906188, e.y);
    return big + small;
}

static void sincos_turns_tab(uint32_t lo, uint32_t hi, double *s, double *c)
{
    const d2 e = TAB[128 + (hi >> 24)];
    const double dl = mk(((hi >> 12) & 0xfffu) | 0x3ff00000u, (hi << 20) | (lo >> 12));  /* 1 + J_low44 / 2^52 */
    const double y = dl - (1.0 + 0x1p-9);
    const double b = fma(y, 6.283185307179586477, 6.283185307179586477 * 0x1p-53);
    const double z = b * b;
    const double ps = fma(z, 1.0 / 120, -1.0 / 6);   // sin b = b + b z ps; the next term is < 1e-17
    const double sb = fma(z * b, ps, b);
    double pc = fma(z, -1.0 / 720, 1.0 / 24);
    pc = fma(z, pc, -0.5);
    const double cm = z * pc;
    *s = fma(e.y, sb, fma(e.x, cm, e.x));
    *c = fma(-e.x, sb, fma(e.y, cm, e.y));
}

static double exp_tab(double x)
{
    const double shifted = fma(x, 369.32993046757462751, 0x1.8p52);   /* 256 / ln 2; integer lands in the low mantissa bits */
    const double n = shifted - 0x1.8p52;
    double r = fma(n, -6.93147180369123816490e-01 / 256, x);
    r = fma(n, -1.90821492927058770002e-10 / 256, r);
    uint32_t shi, slo;
    split(shifted, &shi, &slo);
    const int ni = (int)slo;
    const double T = ((const double *)(TAB + 384))[ni & 255];
    double p = fma(r, 1.0 / 24, 1.0 / 6);
    p = fma(r, p, 0.5);
    p = fma(r, p, 1.0);
    return ldexp(fma(T, r * p, T), ni >> 8);
}

static uint64_t st = 88172645463325252ull;
static uint64_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }

int main(int argc, char **argv)
{
    const long n_iter = argc > 1 ? atol(argv[1]) : 40000000;
    double max_rel = 0, max_abs_sc = 0, max_z = 0;
    for (long it = 0; it < n_iter; ++it) {
        uint64_t a = rnd(), b = rnd();
        if (it < 64) a = ~0ull << it;               /* u -> 1 and the top of each binade */
        else if (it < 128) a = (1ull << (it - 64)); /* tiny u */
        else if (it < 5000) a = (0xb504f333f9de6484ull) + (it - 2500) * 4096; /* around sqrt(1/2) */
        const uint32_t lo = (uint32_t)a, hi = (uint32_t)(a >> 32);
        const double u = u01(lo, hi);
        const double g = neg2log_unit_tab(u);
        const long double ref = -2.0L * logl((long double)u);
        const double rel = (double)fabsl(((long double)g - ref) / ref);
        if (rel > max_rel) max_rel = rel;
        double s, c;
        const uint32_t lo2 = (uint32_t)b, hi2 = (uint32_t)(b >> 32);
        sincos_turns_tab(lo2, hi2, &s, &c);
        const long double ub = (long double)u01(lo2, hi2);
        /* reference with exact octant reduction: angle = 2 pi ub */
        const long double ang = 6.283185307179586476925286766559L * (ub - (ub > 0.5L ? 1.0L : 0.0L));
        const double es = (double)fabsl((long double)s - sinl(ang)), ec = (double)fabsl((long double)c - cosl(ang));
        if (es > max_abs_sc) max_abs_sc = es;
        if (ec > max_abs_sc) max_abs_sc = ec;
        const double zerr = fabs(sqrt(g) * c - (double)(sqrtl(ref) * cosl(ang)));
        if (zerr > max_z) max_z = zerr;
    }
    double max_exp = 0;
    for (long it = 0; it < n_iter / 2; ++it) {
        const double x = ((double)(rnd() >> 11) * 0x1p-53 - 0.5) * (it & 1 ? 80.0 : 4.0);
        const long double ref = expl((long double)x);
        const double rel = (double)fabsl(((long double)exp_tab(x) - ref) / ref);
        if (rel > max_exp) max_exp = rel;
    }
    printf("exp     : max relative error %.3e (%.2f ulp)\n", max_exp, max_exp / 2.22e-16);
    printf("-2 ln u : max relative error %.3e (%.2f ulp)\n", max_rel, max_rel / 2.22e-16);
    printf("sin/cos : max absolute error %.3e\n", max_abs_sc);
    printf("normal  : max absolute error %.3e\n", max_z);
    return !(max_rel < 4e-16 && max_abs_sc < 2e-16 && max_exp < 3e-16 && max_z < 4e-15);
}
