/*
 * legacy_abi.c -- the reference's GPU entry points, by name and signature, on the MI355X engine.
 *
 * Built twice (see Makefile): libmcgpu_f64.so (default) and libmcgpu_f32.so
 * (-DMC_SINGLE_PRECISION), each for one asset count N (-DN=<n>, default 3), exactly like the
 * reference where precision is chosen by directory and N by editing MonteCarlo.h:16.
 *
 * Replaces (marcomatteo/MonteCarloCUDA, double_precision/MonteCarloKernel.cu):
 *   :500 dev_vanillaOpt   :483 dev_basketOpt   :517 dev_cvaEquityOption
 * Behaviour kept from the reference:
 *   - paths simulated = numBlocks * (sims / numBlocks)           (:491,508,524 and :413)
 *   - basket: option->p must already hold the Cholesky factor    (basketOpt.cu:96-99)
 *   - CVA: cva->n = number of exposure dates, cva->ns ignored    (:527)
 *   - a fixed seed: the same call returns the same numbers       (:289)
 *   - any failure prints a message and exits with status 1       (MonteCarlo.h:22-30, :39-46)
 * Behaviour changed: the device context is created on first use and kept for the
 * life of the process (the reference allocates, seeds and frees on every call, :296-363);
 * numThreads is accepted and ignored (it only shaped the reference's reduction, :163);
 * nothing is printed on success.
 * Environment (the one table: INTEGRATION.md section 1; `config()` below resolves them once, at the first call, and MC_VERBOSE=2
 * prints the result): MC_DEVICE, MC_DEVICES (several GPUs: every call sharded and closed by one RCCL all-reduce of the triple,
 * include/mc_multi.h; libmc_multi.so -- and with it RCCL -- is loaded only then), MC_RNG=xorwow (the reference's generator),
 * MC_ANTITHETIC, MC_CONTROL_VARIATE, MC_VERBOSE (1: one line per call plus the stage breakdown of mc_call_stats -- the reference
 * prints its stages from inside every call, dp/MonteCarloKernel.cu:317-323,380-386,404-409,415-427).  Read by EVERY call: MC_SEED and
 * MC_RNG=xorwow_grid (the reference's generator AND its launch geometry: numBlocks and numThreads then shape the sample as they do
 * in the reference -- one XORWOW state per thread seeded blockIdx + gridDim, thread t pricing paths t, t + numThreads, ... of its
 * block; mc_*_run_grid_*; one GPU, MC_SEED is not used).
 */
#define _GNU_SOURCE   /* dladdr */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "MonteCarlo.h"
#include "mc_multi.h"

#ifdef MC_SINGLE_PRECISION
#define API(name) name##_f32
typedef mc_option_f32 api_option;
typedef mc_basket_f32 api_basket;
typedef mc_cva_f32 api_cva;
#else
#define API(name) name##_f64
typedef mc_option_f64 api_option;
typedef mc_basket_f64 api_basket;
typedef mc_cva_f64 api_cva;
#endif

static mc_context *g_ctx;

/* ---- several GPUs (MC_DEVICES): libmc_multi.so, loaded on demand from this library's own directory ---- */
static mc_multi *g_multi;
static const char *(*multi_last_error)(void);
static void (*multi_destroy)(mc_multi *);
typedef int (*multi_run_fn)(mc_multi *, const void *, uint64_t, uint64_t, uint64_t, mc_result *);
static multi_run_fn multi_vanilla, multi_basket, multi_cva;

static void die(const char *what)
{
    const char *text = mc_last_error();
    if (multi_last_error && multi_last_error()[0])
        text = multi_last_error();
    fprintf(stderr, "Error %s: %s\n", what, text);
    exit(1);
}

/* ---- configuration: every environment variable this library reads, resolved ONCE at the first call (MC_SEED alone is read by
 * every call) and printed on request (MC_VERBOSE=2).  INTEGRATION.md section 1 holds the table. ---- */
typedef struct {
    int verbose;           /* MC_VERBOSE: 1 = one line per call with the stage breakdown, 2 = also the resolved configuration */
    int device;            /* MC_DEVICE */
    const char *devices;   /* MC_DEVICES: NULL / "" = single device */
    int antithetic;        /* MC_ANTITHETIC */
    int control;           /* MC_CONTROL_VARIATE */
    int xorwow;            /* MC_RNG = xorwow: the generator of the context (xorwow_grid is a property of each CALL: grid_mode()) */
} Config;

static const Config *config(void)
{
    static Config c;
    static int resolved;
    if (resolved)
        return &c;
    const char *v;
    c.verbose = (v = getenv("MC_VERBOSE")) ? (atoi(v) > 0 ? atoi(v) : 1) : 0;   /* set at all = 1, as before; a number selects the level */
    c.device = (v = getenv("MC_DEVICE")) ? atoi(v) : 0;
    c.devices = (v = getenv("MC_DEVICES")) && v[0] ? v : NULL;
    c.antithetic = (v = getenv("MC_ANTITHETIC")) && atoi(v);
    c.control = (v = getenv("MC_CONTROL_VARIATE")) && atoi(v);
    c.xorwow = (v = getenv("MC_RNG")) && !strcmp(v, "xorwow");
    resolved = 1;
    if (c.verbose >= 2)
        fprintf(stderr, "legacy symbols config (%s, N=%d): MC_DEVICE=%d MC_DEVICES=%s MC_RNG=%s MC_ANTITHETIC=%d MC_CONTROL_VARIATE=%d MC_SEED=%s MC_VERBOSE=%d\n",
#ifdef MC_SINGLE_PRECISION
                "libmcgpu_f32",
#else
                "libmcgpu_f64",
#endif
                N, c.device, c.devices ? c.devices : "(unset: one device)", getenv("MC_RNG") ? getenv("MC_RNG") : "(unset: philox)", c.antithetic,
                c.control, getenv("MC_SEED") ? getenv("MC_SEED") : "(unset: MC_DEFAULT_SEED)", c.verbose);
    return &c;
}

static void drop_multi(void)
{
    if (g_multi)
        multi_destroy(g_multi);
    g_multi = NULL;
}

static void *multi_symbol(void *lib, const char *name)
{
    void *p = dlsym(lib, name);
    if (!p) {
        fprintf(stderr, "Error: libmc_multi.so lacks %s\n", name);
        exit(1);
    }
    return p;
}

/* NULL unless MC_DEVICES is set */
static mc_multi *multi(void)
{
    const char *list = config()->devices;
    if (g_multi || !list)
        return g_multi;
    char path[4096] = "libmc_multi.so";
    Dl_info here;
    if (dladdr((void *)&multi, &here) && here.dli_fname) {   /* next to this library, wherever it was installed */
        const char *slash = strrchr(here.dli_fname, '/');
        if (slash && (size_t)(slash - here.dli_fname) + 16 < sizeof path) {
            memcpy(path, here.dli_fname, (size_t)(slash - here.dli_fname) + 1);
            strcpy(path + (slash - here.dli_fname) + 1, "libmc_multi.so");
        }
    }
    void *lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        fprintf(stderr, "Error: MC_DEVICES is set but %s cannot be loaded: %s\n", path, dlerror());
        exit(1);
    }
    int (*create)(const int *, int, int, mc_multi **) = (int (*)(const int *, int, int, mc_multi **))multi_symbol(lib, "mc_multi_create");
    int (*set_anti)(mc_multi *, int) = (int (*)(mc_multi *, int))multi_symbol(lib, "mc_multi_set_antithetic");
    int (*set_cv)(mc_multi *, int) = (int (*)(mc_multi *, int))multi_symbol(lib, "mc_multi_set_control_variate");
    int (*set_rng)(mc_multi *, int, uint64_t) = (int (*)(mc_multi *, int, uint64_t))multi_symbol(lib, "mc_multi_set_generator");
    int (*set_timing)(mc_multi *, int) = (int (*)(mc_multi *, int))multi_symbol(lib, "mc_multi_set_timing");
    multi_last_error = (const char *(*)(void))multi_symbol(lib, "mc_multi_last_error");
    multi_destroy = (void (*)(mc_multi *))multi_symbol(lib, "mc_multi_destroy");
#ifdef MC_SINGLE_PRECISION
    multi_vanilla = (multi_run_fn)multi_symbol(lib, "mc_multi_vanilla_run_f32");
    multi_basket = (multi_run_fn)multi_symbol(lib, "mc_multi_basket_run_f32");
    multi_cva = (multi_run_fn)multi_symbol(lib, "mc_multi_cva_run_f32");
#else
    multi_vanilla = (multi_run_fn)multi_symbol(lib, "mc_multi_vanilla_run_f64");
    multi_basket = (multi_run_fn)multi_symbol(lib, "mc_multi_basket_run_f64");
    multi_cva = (multi_run_fn)multi_symbol(lib, "mc_multi_cva_run_f64");
#endif
    int devices[64], n = 0;
    if (strcmp(list, "all") != 0) {
        for (const char *p = list; *p && n < 64;) {
            char *end;
            const long d = strtol(p, &end, 10);
            if (end == p) {
                fprintf(stderr, "Error: MC_DEVICES=\"%s\" is not a comma-separated list of device numbers (or \"all\")\n", list);
                exit(1);
            }
            devices[n++] = (int)d;
            p = *end == ',' ? end + 1 : end;
        }
    }
    if (create(n ? devices : NULL, n, 0, &g_multi) != MC_OK)
        die("creating the multi-device handle (MC_DEVICES)");
    if (config()->antithetic)
        set_anti(g_multi, 1);
    if (config()->control)
        set_cv(g_multi, 1);
    if (config()->xorwow)
        set_rng(g_multi, MC_RNG_XORWOW, 0);
    if (!config()->verbose)
        set_timing(g_multi, 0);
    atexit(drop_multi);
    return g_multi;
}

static void drop_context(void)
{
    mc_context_destroy(g_ctx);
    g_ctx = NULL;
}

static mc_context *context(void)
{
    if (!g_ctx) {
        if (mc_context_create(config()->device, 0, &g_ctx) != MC_OK)
            die("creating the device context");
        if (config()->antithetic)
            mc_context_set_antithetic(g_ctx, 1);
        if (config()->control)
            mc_context_set_control_variate(g_ctx, 1);
        if (config()->xorwow)
            mc_context_set_generator(g_ctx, MC_RNG_XORWOW, 0);
        if (!config()->verbose)   /* nobody reads kernel_ms: take the short way back (result polled from pinned memory) */
            mc_context_set_timing(g_ctx, 0);
        atexit(drop_context);
    }
    return g_ctx;
}

/* MC_RNG=xorwow_grid: the calls go through mc_*_run_grid_* with the caller's (numBlocks, numThreads) */
static int grid_mode(void)
{
    const char *r = getenv("MC_RNG");   /* read by every call, like MC_SEED: a process may price both ways */
    if (!r || strcmp(r, "xorwow_grid"))
        return 0;
    if (config()->devices) {
        fprintf(stderr, "Error: MC_RNG=xorwow_grid reproduces a single-GPU launch of the reference; unset MC_DEVICES\n");
        exit(1);
    }
    return 1;
}

/* the seed of the next call: the *_ex entry points set it for their own call, everything else reads MC_SEED */
static int g_seed_given;
static uint64_t g_seed;
static uint64_t seed(void)
{
    if (g_seed_given)
        return g_seed;
    const char *s = getenv("MC_SEED");
    return s ? strtoull(s, NULL, 0) : MC_DEFAULT_SEED;
}

/* the reference's path-count rule; also guards the divisions it leaves unguarded */
static uint64_t path_count(int numBlocks, int sims)
{
    if (numBlocks <= 0 || sims <= 0 || sims / numBlocks <= 0) {
        fprintf(stderr, "Error: numBlocks=%d, sims=%d give no paths\n", numBlocks, sims);
        exit(1);
    }
    return (uint64_t)numBlocks * (uint64_t)(sims / numBlocks);
}

static OptionValue finish(const mc_result *r, const char *what)
{
    OptionValue v;
    v.Expected = (mc_real)r->expected;
    v.Confidence = (mc_real)r->confidence;
    if (config()->verbose) {
        /* the reference prints its stages from inside every call (RNG set-up :317-323, allocations :326-341, kernel :380-386, copy
         * :404-409, closing :415-427); the same breakdown for this call, and what the first call of the process paid before it */
        printf("%s: %llu paths, kernel %.3f ms, call %.3f ms, value %.9g +- %.3g\n", what, (unsigned long long)r->n,
               r->kernel_ms, r->wall_ms, r->expected, r->confidence);
        mc_call_stats k;
        if (!g_multi && g_ctx && mc_context_last_call_stats(g_ctx, &k) == MC_OK) {
            printf("%s stages (ms): set-up %.3f | tables %.3f | launch %.3f | kernel %.3f | read-back %.3f | closing %.3f | = call %.3f", what,
                   k.setup_ms, k.table_upload_ms, k.launch_ms, k.kernel_ms, k.readback_ms, k.closing_ms, k.wall_ms);
            if (k.first_call)
                printf("   [first call of this context: creating it (HIP runtime, stream, buffers) took %.1f ms before that]", k.context_create_ms);
            printf("\n");
        }
    }
    return v;
}

OptionValue dev_vanillaOpt(OptionData *opt, int numBlocks, int numThreads, int sims)
{
    (void)numThreads;   /* shapes the sample only under MC_RNG=xorwow_grid */
    const api_option o = {opt->s, opt->k, opt->r, opt->v, opt->t};
    mc_result r;
    const uint64_t n = path_count(numBlocks, sims);
    if (grid_mode()) {
        if (API(mc_vanilla_run_grid)(context(), &o, numBlocks, numThreads, n / (uint64_t)numBlocks, &r) != MC_OK)
            die("in dev_vanillaOpt");
        return finish(&r, "dev_vanillaOpt");
    }
    if ((multi() ? multi_vanilla(g_multi, &o, seed(), 0, n, &r) : API(mc_vanilla_run)(context(), &o, seed(), 0, n, &r)) != MC_OK)
        die("in dev_vanillaOpt");
    return finish(&r, "dev_vanillaOpt");
}

OptionValue dev_basketOpt(MultiOptionData *option, int numBlocks, int numThreads, int sims)
{
    (void)numThreads;   /* shapes the sample only under MC_RNG=xorwow_grid */
    const api_basket b = {N, option->s, option->v, &option->p[0][0], option->d, option->w,
                          option->k, option->t, option->r};
    mc_result r;
    const uint64_t n = path_count(numBlocks, sims);
    if (grid_mode()) {
        if (API(mc_basket_run_grid)(context(), &b, numBlocks, numThreads, n / (uint64_t)numBlocks, &r) != MC_OK)
            die("in dev_basketOpt");
        return finish(&r, "dev_basketOpt");
    }
    if ((multi() ? multi_basket(g_multi, &b, seed(), 0, n, &r) : API(mc_basket_run)(context(), &b, seed(), 0, n, &r)) != MC_OK)
        die("in dev_basketOpt");
    return finish(&r, "dev_basketOpt");
}

OptionValue dev_cvaEquityOption(CVA *cva, int numBlocks, int numThreads, int sims)
{
    (void)numThreads;   /* shapes the sample only under MC_RNG=xorwow_grid */
    const api_cva c = {cva->defInt, cva->lgd,
                       {cva->option.s, cva->option.k, cva->option.r, cva->option.v, cva->option.t}, cva->n};
    mc_result r;
    const uint64_t n = path_count(numBlocks, sims);
    if (grid_mode()) {
        if (API(mc_cva_run_grid)(context(), &c, numBlocks, numThreads, n / (uint64_t)numBlocks, &r) != MC_OK)
            die("in dev_cvaEquityOption");
        return finish(&r, "dev_cvaEquityOption");
    }
    if ((multi() ? multi_cva(g_multi, &c, seed(), 0, n, &r) : API(mc_cva_run)(context(), &c, seed(), 0, n, &r)) != MC_OK)
        die("in dev_cvaEquityOption");
    return finish(&r, "dev_cvaEquityOption");
}

/* ---- explicit-seed variants (not in the reference, whose API has no seed parameter: SURVEY 8b "RNG contract").
 * Same calls with `seed` in place of MC_SEED / the default; under MC_RNG=xorwow_grid the seed is the geometry's and
 * `seed` is not used.  Not re-entrant, like the reference's entry points (process-global state). ---- */
OptionValue dev_vanillaOpt_ex(OptionData *opt, int numBlocks, int numThreads, int sims, uint64_t seed_)
{
    g_seed_given = 1, g_seed = seed_;
    const OptionValue v = dev_vanillaOpt(opt, numBlocks, numThreads, sims);
    g_seed_given = 0;
    return v;
}

OptionValue dev_basketOpt_ex(MultiOptionData *option, int numBlocks, int numThreads, int sims, uint64_t seed_)
{
    g_seed_given = 1, g_seed = seed_;
    const OptionValue v = dev_basketOpt(option, numBlocks, numThreads, sims);
    g_seed_given = 0;
    return v;
}

OptionValue dev_cvaEquityOption_ex(CVA *cva, int numBlocks, int numThreads, int sims, uint64_t seed_)
{
    g_seed_given = 1, g_seed = seed_;
    const OptionValue v = dev_cvaEquityOption(cva, numBlocks, numThreads, sims);
    g_seed_given = 0;
    return v;
}
