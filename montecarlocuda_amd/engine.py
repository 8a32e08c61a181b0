"""Host-side mirror of the reference's GPU interface on top of the C ABI (ctypes, no torch).

Names follow the reference (marcomatteo/MonteCarloCUDA): ``OptionData``, ``MultiOptionData``,
``CVA``, ``OptionValue`` (MonteCarlo.h:32-65) and ``dev_vanillaOpt`` / ``dev_basketOpt`` /
``dev_cvaEquityOption`` (MonteCarloKernel.cu:500,483,517).  ``Engine`` is the persistent
per-GPU context underneath, with the explicit-seed / path-range API the multi-GPU harness uses.
The simulation itself always runs in libmc_mi355x.so on the GPU; nothing here computes paths.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import MC_DEFAULT_SEED, check, lib

NP = {"f32": np.float32, "f64": np.float64}


# ---- reference-shaped inputs / outputs ----------------------------------------------------
@dataclass
class OptionData:           # MonteCarlo.h:32-38
    s: float
    k: float
    r: float
    v: float
    t: float


@dataclass
class MultiOptionData:      # MonteCarlo.h:41-50; p = Cholesky factor at call time
    s: Sequence[float]
    v: Sequence[float]
    p: Sequence[Sequence[float]]
    d: Sequence[float]
    w: Sequence[float]
    k: float
    t: float
    r: float


@dataclass
class CVA:                  # MonteCarlo.h:57-65
    defInt: float
    lgd: float
    option: OptionData
    n: int
    ns: int = 0


@dataclass
class OptionValue:          # MonteCarlo.h:52-55
    Expected: float
    Confidence: float


@dataclass
class Estimate:
    expected: float
    confidence: float
    sum: float
    sum2: float
    n: int
    kernel_ms: float = 0.0
    wall_ms: float = 0.0

    def value(self) -> OptionValue:
        return OptionValue(self.expected, self.confidence)


def _as_option(X, o) -> C.Structure:
    if isinstance(o, dict):
        o = OptionData(**{k: o[k] for k in "skrvt"})
    return _lib.OPTION[X](o.s, o.k, o.r, o.v, o.t)


class _BasketHolder:
    """Keeps the numpy arrays alive for as long as the ctypes struct is in use."""

    def __init__(self, X, b):
        if isinstance(b, dict):
            b = MultiOptionData(**{k: b[k] for k in ("s", "v", "p", "d", "w", "k", "t", "r")})
        dt = NP[X]
        self.n = len(b.s)
        self.arrs = [np.ascontiguousarray(a, dtype=dt) for a in (b.s, b.v, np.asarray(b.p, dtype=dt).reshape(-1),
                                                                  b.d, b.w)]
        if self.arrs[2].size != self.n * self.n or any(a.size != self.n for a in (self.arrs[0], self.arrs[1],
                                                                               self.arrs[3], self.arrs[4])):
            raise ValueError("basket arrays disagree on n")
        P = C.POINTER(_lib.CT[X])
        self.struct = _lib.BASKET[X](self.n, *[a.ctypes.data_as(P) for a in self.arrs], b.k, b.t, b.r)


def _as_cva(X, c) -> C.Structure:
    if isinstance(c, dict):
        c = CVA(c["defint"], c["lgd"], OptionData(*(c[k] for k in "skrvt")), c["n_grid"])
    return _lib.CVA[X](c.defInt, c.lgd, _as_option(X, c.option), c.n)


def _estimate(r: _lib.Result) -> Estimate:
    return Estimate(r.expected, r.confidence, r.sum, r.sum2, int(r.n), float(r.kernel_ms), float(r.wall_ms))


class Engine:
    """One persistent context on one MI355X (mc_context_create / mc_context_destroy)."""

    def __init__(self, device: int = 0, blocks: int = 0):
        self._ctx = C.c_void_p()
        check(lib().mc_context_create(device, blocks, C.byref(self._ctx)))
        self.device = device
        self.blocks = lib().mc_context_blocks(self._ctx)
        self.stream = lib().mc_context_stream(self._ctx) or 0
        # the C context also enters the fp32-normals mode from the environment at creation (mc_api.hip: context_allocate)
        self._normals_f32 = os.environ.get("MC_F64_NORMALS") == "f32"

    def close(self):
        if self._ctx:
            lib().mc_context_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        name = C.create_string_buffer(128)
        cus, mhz = C.c_int(), C.c_int()
        check(lib().mc_context_info(self._ctx, name, 128, C.byref(cus), C.byref(mhz)))
        return {"name": name.value.decode(), "compute_units": cus.value, "clock_mhz": mhz.value,
                "blocks": self.blocks}

    def last_call_stats(self):
        """Stage breakdown (ms) of this context's last synchronous call: mc_context_last_call_stats."""
        from ._lib import CallStats
        k = CallStats()
        check(lib().mc_context_last_call_stats(self._ctx, C.byref(k)))
        return {f: getattr(k, f) for f, _ in CallStats._fields_}

    def describe(self):
        """The context's resolved configuration as one line (what MC_VERBOSE=2 prints at creation)."""
        buf = C.create_string_buffer(1024)
        check(lib().mc_context_describe(self._ctx, buf, 1024))
        return buf.value.decode()

    def last_launch(self):
        """(workgroups, lanes per workgroup) of the most recent simulation launch of this context."""
        g, t = C.c_int(), C.c_int()
        check(lib().mc_context_last_launch(self._ctx, C.byref(g), C.byref(t)))
        return g.value, t.value

    def set_antithetic(self, on: bool):
        """Plain Monte Carlo (the reference's estimator) or antithetic variates (pair means)."""
        check(lib().mc_context_set_antithetic(self._ctx, 1 if on else 0))

    def set_control_variate(self, on: bool):
        """Baskets: simulate payoff(arithmetic) - payoff(geometric) and add the geometric closed form back."""
        check(lib().mc_context_set_control_variate(self._ctx, 1 if on else 0))

    def order(self, stream: int):
        """Make `stream` (a hipStream_t handle) wait for everything this context has enqueued so far."""
        check(lib().mc_context_order(self._ctx, C.c_void_p(stream)))

    def idle(self) -> bool:
        """True when everything this context has enqueued has completed (a user-space poll, no sleeping)."""
        r = lib().mc_context_idle(self._ctx)
        if r < 0:
            raise _lib.McError("mc_context_idle failed")
        return r == 1

    def arm_direct(self):
        """The NEXT launch() also delivers its triple into a pinned host slot (mc_context_arm_direct).  Returns the slot
        (ctypes pointer to 3 doubles, word 2 preset to -1): poll with wait_slot()."""
        slot = C.POINTER(C.c_double)()
        check(lib().mc_context_arm_direct(self._ctx, C.byref(slot)))
        return slot

    def publish(self, d_src_ptr: int, stream: int):
        """Enqueue on `stream` the copy of the 3 doubles at device address d_src_ptr into a pinned host slot
        (mc_context_publish): how a triple that another kernel produced -- an all-reduce -- reaches a polling host."""
        slot = C.POINTER(C.c_double)()
        check(lib().mc_context_publish(self._ctx, C.c_void_p(d_src_ptr), C.c_void_p(stream), C.byref(slot)))
        return slot

    @staticmethod
    def wait_slot(slot, timeout_s: float = 30.0):
        """Poll a slot from user space until its n word is written; returns (sum, sum2, n)."""
        import time
        t0 = time.perf_counter()
        spins = 0
        while slot[2] == -1.0:
            spins += 1
            if (spins & 0xFFFF) == 0 and time.perf_counter() - t0 > timeout_s:
                raise _lib.McError("the device never delivered the result")
        return slot[0], slot[1], slot[2]

    def set_timing(self, on: bool):
        """Synchronous calls: HIP events + copy + synchronize (kernel_ms reported; default) or, off, the result written
        straight to pinned host memory and polled from user space (kernel_ms = 0, ~10 us less per call)."""
        check(lib().mc_context_set_timing(self._ctx, 1 if on else 0))

    def set_finish(self, fused: bool):
        """Final reduction inside the simulation kernel (default) or as a second launch (A/B baseline)."""
        check(lib().mc_context_set_finish(self._ctx, 1 if fused else 0))

    def set_generator(self, name: str, subsequence_base: int = 0):
        """'philox' (default; counter-based) or 'xorwow' (the reference's generator, one sequence per lane)."""
        check(lib().mc_context_set_generator(self._ctx, {"philox": 0, "xorwow": 1}[name], subsequence_base))

    def set_cva_date_lanes(self, lanes: int):
        """CVA: adjacent lanes sharing one path's dates (0 = by call size, 1 = never, 2 ... 64 = the whole call date-parallel)."""
        check(lib().mc_context_set_cva_date_lanes(self._ctx, int(lanes)))

    def set_normals(self, mode: str):
        """fp64 kernels: 'native' (default; true fp64 normals) or 'f32' (the reference's dp arithmetic: a float normal
        widened to double, four per Philox block)."""
        check(lib().mc_context_set_normals(self._ctx, {"native": 0, "f32": 1}[mode]))
        self._normals_f32 = mode == "f32"

    def xorwow_words(self, seed, first_subsequence, n_subsequences, words_each):
        out = np.empty((n_subsequences, words_each), dtype=np.uint32)
        check(lib().mc_xorwow_words(self._ctx, seed, first_subsequence, n_subsequences, words_each,
                                    out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def profile(self, every: int):
        """Sample the simulation kernel's device time on every `every`-th launch (0 = off)."""
        check(lib().mc_context_profile(self._ctx, every))

    def profile_read(self):
        """(samples, total_ms) of the sampled simulation-kernel launches since the last read."""
        n, ms = C.c_int(), C.c_double()
        check(lib().mc_context_profile_read(self._ctx, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    # ---- synchronous estimators --------------------------------------------------------
    def _run(self, prod, X, struct, seed, first, n):
        r = _lib.Result()
        check(getattr(lib(), f"mc_{prod}_run_{X}")(self._ctx, C.byref(struct), seed, first, n, C.byref(r)))
        return _estimate(r)

    def vanilla(self, opt, n_paths, seed=MC_DEFAULT_SEED, first_path=0, precision="f64") -> Estimate:
        return self._run("vanilla", precision, _as_option(precision, opt), seed, first_path, n_paths)

    def vanilla_greeks(self, opt, n_paths, seed=MC_DEFAULT_SEED, first_path=0, precision="f64"):
        """(price, delta, vega) Estimates from one pass (pathwise derivatives)."""
        g = _lib.Greeks()
        check(getattr(lib(), f"mc_vanilla_greeks_run_{precision}")(self._ctx, C.byref(_as_option(precision, opt)), seed,
                                                                    first_path, n_paths, C.byref(g)))
        return _estimate(g.price), _estimate(g.delta), _estimate(g.vega)

    def vanilla_greeks_lr(self, opt, n_paths, seed=MC_DEFAULT_SEED, first_path=0, precision="f64"):
        """(price, delta, vega) Estimates by the likelihood-ratio method (score of the density x payoff)."""
        g = _lib.Greeks()
        check(getattr(lib(), f"mc_vanilla_greeks_lr_run_{precision}")(self._ctx, C.byref(_as_option(precision, opt)), seed,
                                                                       first_path, n_paths, C.byref(g)))
        return _estimate(g.price), _estimate(g.delta), _estimate(g.vega)

    def basket_greeks(self, b, n_paths, seed=MC_DEFAULT_SEED, first_path=0, precision="f64", lr=False):
        """(price, [delta per asset], [vega per asset]) Estimates: pathwise derivatives, or (lr=True) likelihood-ratio forms."""
        h = _BasketHolder(precision, b)
        price = _lib.Result()
        delta, vega = (_lib.Result * h.n)(), (_lib.Result * h.n)()
        check(getattr(lib(), f"mc_basket_greeks{'_lr' if lr else ''}_run_{precision}")(self._ctx, C.byref(h.struct), seed, first_path, n_paths,
                                                                                        C.byref(price), delta, vega))
        return _estimate(price), [_estimate(x) for x in delta], [_estimate(x) for x in vega]

    def cva_greeks(self, c, n_paths, seed=MC_DEFAULT_SEED, first_path=0, precision="f64", lr=False):
        """(cva, delta, vega) Estimates: the CVA and its derivatives with respect to spot and volatility, pathwise or (lr=True)
        by the likelihood ratio."""
        g = _lib.CvaGreeks()
        check(getattr(lib(), f"mc_cva_greeks{'_lr' if lr else ''}_run_{precision}")(self._ctx, C.byref(_as_cva(precision, c)), seed, first_path,
                                                                                     n_paths, C.byref(g)))
        return _estimate(g.cva), _estimate(g.delta), _estimate(g.vega)

    def basket(self, b, n_paths, seed=MC_DEFAULT_SEED, first_path=0, precision="f64") -> Estimate:
        h = _BasketHolder(precision, b)
        return self._run("basket", precision, h.struct, seed, first_path, n_paths)

    def cva(self, c, n_paths, seed=MC_DEFAULT_SEED, first_path=0, precision="f64") -> Estimate:
        return self._run("cva", precision, _as_cva(precision, c), seed, first_path, n_paths)

    # ---- asynchronous launches (device triple, caller's stream) ------------------------
    def launch(self, prod, precision, struct, seed, first_path, n_paths, d_triple_ptr: int, stream: int = 0):
        """Enqueue; d_triple_ptr = device address of 3 doubles, stream = hipStream_t handle (0 = the HIP null
        stream; ``self.stream`` is the context's own)."""
        check(getattr(lib(), f"mc_{prod}_launch_{precision}")(self._ctx, C.byref(struct), seed, first_path,
                                                               n_paths, C.c_void_p(d_triple_ptr),
                                                               C.c_void_p(stream)))

    def prepared(self, prod, precision, inputs):
        """ctypes struct (plus whatever must stay alive) for repeated launch() calls."""
        if prod == "vanilla":
            return _as_option(precision, inputs), None
        if prod == "basket":
            h = _BasketHolder(precision, inputs)
            return h.struct, h
        return _as_cva(precision, inputs), None

    # ---- per-path values (parity tests) ------------------------------------------------
    def _paths(self, prod, X, struct, seed, first, n):
        out = np.empty(n, dtype=NP[X])
        check(getattr(lib(), f"mc_{prod}_paths_{X}")(self._ctx, C.byref(struct), seed, first, n,
                                                      out.ctypes.data_as(C.POINTER(_lib.CT[X]))))
        return out

    def vanilla_paths(self, opt, n_paths, seed=MC_DEFAULT_SEED, first_path=0, precision="f64"):
        return self._paths("vanilla", precision, _as_option(precision, opt), seed, first_path, n_paths)

    def basket_paths(self, b, n_paths, seed=MC_DEFAULT_SEED, first_path=0, precision="f64"):
        h = _BasketHolder(precision, b)
        return self._paths("basket", precision, h.struct, seed, first_path, n_paths)

    def cva_paths(self, c, n_paths, seed=MC_DEFAULT_SEED, first_path=0, precision="f64"):
        return self._paths("cva", precision, _as_cva(precision, c), seed, first_path, n_paths)

    def normals(self, seed, domain, first_unit, n_units, block=0, precision="f64"):
        npb = 4 if (precision == "f32" or self._normals_f32) else 8
        out = np.empty(n_units * npb, dtype=NP[precision])
        check(getattr(lib(), f"mc_normals_{precision}")(self._ctx, seed, domain, first_unit, n_units, block,
                                                         out.ctypes.data_as(C.POINTER(_lib.CT[precision]))))
        return out.reshape(n_units, npb)

    # ---- compatibility mode: the reference's launch geometry and per-thread XORWOW streams (mc_*_run_grid_*) ----
    def run_grid(self, prod, inputs, num_blocks, num_threads, paths_per_block, precision="f64") -> Estimate:
        """num_blocks * paths_per_block paths drawn the reference's way (dp/MonteCarloKernel.cu:285-290: one XORWOW state per
        thread, seed blockIdx + gridDim, subsequence threadIdx; thread t prices paths t, t + T, ... of its block)."""
        struct, keep = self.prepared(prod, precision, inputs)
        r = _lib.Result()
        check(getattr(lib(), f"mc_{prod}_run_grid_{precision}")(self._ctx, C.byref(struct), num_blocks, num_threads, paths_per_block,
                                                                C.byref(r)))
        return _estimate(r)

    def set_grid_form(self, form: str):
        """Launch-geometry mode: 'auto' (fused kernels where they exist), 'staged' (normals through HBM: the checker) or
        'fused' (test hook, mc_mi355x_test.h)."""
        check(lib().mc_context_set_grid_form(self._ctx, _lib.GRID_FORM[form]))

    def paths_grid(self, prod, inputs, num_blocks, num_threads, paths_per_block, precision="f64"):
        """Per-path values of a launch-geometry call in the call's path order, (num_blocks, paths_per_block) (test hook)."""
        struct, keep = self.prepared(prod, precision, inputs)
        out = np.empty((num_blocks, paths_per_block), dtype=NP[precision])
        check(getattr(lib(), f"mc_{prod}_paths_grid_{precision}")(self._ctx, C.byref(struct), num_blocks, num_threads, paths_per_block,
                                                                  out.ctypes.data_as(C.POINTER(_lib.CT[precision]))))
        return out

    def grid_normals(self, num_blocks, num_threads, count):
        """The first `count` normals of every thread's stream: (num_blocks, num_threads, count) float32."""
        out = np.empty((num_blocks, num_threads, count), dtype=np.float32)
        check(lib().mc_grid_normals(self._ctx, num_blocks, num_threads, count, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    # ---- test hooks: the simulation kernels on a caller-supplied normal stream -----------------------
    def _from_normals(self, prod, X, struct, normals, n_paths, flags, want_values):
        z = np.ascontiguousarray(normals, dtype=NP[X]).reshape(-1)
        P = C.POINTER(_lib.CT[X])
        vals = np.empty(n_paths, dtype=NP[X]) if want_values else None
        r = _lib.Result()
        f = getattr(lib(), f"mc_{prod}_from_normals_{X}")
        args = [self._ctx, C.byref(struct), z.ctypes.data_as(P), n_paths]
        if prod != "vanilla":
            args.append(flags)
        args += [vals.ctypes.data_as(P) if want_values else None, C.byref(r)]
        check(f(*args))
        return _estimate(r), vals

    def vanilla_from_normals(self, opt, normals, precision="f64", want_values=True):
        """(Estimate, per-path payoffs): the vanilla hot kernel with normals[i] as path i's normal."""
        z = np.asarray(normals).reshape(-1)
        return self._from_normals("vanilla", precision, _as_option(precision, opt), z, len(z), 0, want_values)

    def basket_from_normals(self, b, normals, precision="f64", no_vol=False, want_values=True):
        """normals[i, a]: path i, asset a, in drawing order.  no_vol: the reference dp CPU path's model (goldens only)."""
        h = _BasketHolder(precision, b)
        z = np.asarray(normals).reshape(-1, h.n)
        return self._from_normals("basket", precision, h.struct, z, z.shape[0], _lib.FROM_NORMALS_NO_VOL if no_vol else 0, want_values)

    def cva_from_normals(self, c, normals, precision="f64", host_order=False, want_values=True):
        """normals[i, j - 1]: path i, date j.  host_order: the reference CPU loop's lagged exposure (goldens only)."""
        st = _as_cva(precision, c)
        z = np.asarray(normals).reshape(-1, st.n_grid)
        return self._from_normals("cva", precision, st, z, z.shape[0], _lib.FROM_NORMALS_HOST_ORDER if host_order else 0, want_values)


# ---- host helpers (no GPU needed) -----------------------------------------------------------
def closing(sum_, sum2, n, discount=1.0):
    e, c = C.c_double(), C.c_double()
    lib().mc_closing(sum_, sum2, n, discount, C.byref(e), C.byref(c))
    return e.value, c.value


def pci_bus_id(device):
    """PCI bus id of a visible device ("0000:75:00.0"): which physical GPU an index is."""
    buf = C.create_string_buffer(32)
    check(lib().mc_device_pci_bus_id(device, buf, 32))
    return buf.value.decode()


def shard_range(total, rank, world):
    first, count = C.c_uint64(), C.c_uint64()
    lib().mc_shard_range(total, rank, world, C.byref(first), C.byref(count))
    return first.value, count.value


def basket_control_mean(b, precision="f64"):
    """Closed-form E[max(G - K, 0)] of the geometric-basket control (undiscounted, fp64)."""
    h = _BasketHolder(precision, b)
    m = C.c_double()
    check(getattr(lib(), f"mc_basket_control_mean_{precision}")(C.byref(h.struct), C.byref(m)))
    return m.value


def chol(c, precision="f64"):
    """Cholesky with the reference's semantics (MonteCarloHost.c:90-105).  Returns (factor, bad_pivots)."""
    c = np.ascontiguousarray(c, dtype=NP[precision])
    n = c.shape[0]
    a = np.zeros_like(c)
    P = C.POINTER(_lib.CT[precision])
    bad = getattr(lib(), f"mc_chol_{precision}")(n, c.ctypes.data_as(P), a.ctypes.data_as(P))
    return a, bad


def factor_from_cov(cov, precision="f64"):
    """Volatilities and Cholesky factor of the correlation matrix from a covariance matrix (mc_factor_from_cov_*).
    Returns (v, factor, bad_pivots); raises ValueError for a non-positive or non-finite diagonal."""
    cov = np.ascontiguousarray(cov, dtype=NP[precision])
    n = cov.shape[0]
    v = np.zeros(n, dtype=NP[precision])
    p = np.zeros((n, n), dtype=NP[precision])
    P = C.POINTER(_lib.CT[precision])
    bad = getattr(lib(), f"mc_factor_from_cov_{precision}")(n, cov.ctypes.data_as(P), v.ctypes.data_as(P), p.ctypes.data_as(P))
    if bad < 0:
        raise ValueError("covariance matrix needs a finite, positive diagonal and finite entries")
    return v, p, bad


# ---- the reference's three entry points ------------------------------------------------------
_default_engine: Optional[Engine] = None


def default_engine() -> Engine:
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


def _path_count(numBlocks, sims):
    if numBlocks <= 0 or sims // numBlocks <= 0:
        raise ValueError("numBlocks / sims give no paths")
    return numBlocks * (sims // numBlocks)   # MonteCarloKernel.cu:491,508,524 with :413


def _dev(prod, inputs, numBlocks, numThreads, sims, precision) -> OptionValue:
    n = _path_count(numBlocks, sims)
    if os.environ.get("MC_RNG") == "xorwow_grid":   # as the legacy symbols: the reference's launch geometry shapes the sample
        return default_engine().run_grid(prod, inputs, numBlocks, numThreads, n // numBlocks, precision).value()
    return getattr(default_engine(), prod)(inputs, n, precision=precision).value()


def dev_vanillaOpt(opt, numBlocks, numThreads, sims, precision="f64") -> OptionValue:
    return _dev("vanilla", opt, numBlocks, numThreads, sims, precision)


def dev_basketOpt(option, numBlocks, numThreads, sims, precision="f64") -> OptionValue:
    return _dev("basket", option, numBlocks, numThreads, sims, precision)


def dev_cvaEquityOption(cva, numBlocks, numThreads, sims, precision="f64") -> OptionValue:
    return _dev("cva", cva, numBlocks, numThreads, sims, precision)
