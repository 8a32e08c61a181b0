"""Where a call's time goes (mc_call_stats, include/mc_mi355x.h) -- the reference prints these stages from inside every call
(RNG set-up dp/MonteCarloKernel.cu:317-323, allocations :326-341, kernel :380-386, copy :404-409, closing :415-427; its drivers
time around the whole dev_* call, dp/vanillaOpt.cu:77-83); here they are returned per call and printed by the legacy symbols
under MC_VERBOSE=1.  Checked: the parts add up to wall_ms, set-up work is reported as set-up (and is no longer inside
kernel_ms), a repeated call has none, and the verbose modes print what they promise."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAN = dict(s=100.0, k=100.0, r=0.048790, v=0.2, t=1.0)
CVA = dict(s=100.0, k=100.0, r=0.05, v=0.2, t=1.0, defint=0.03, lgd=0.6, n_grid=256)
PARTS = ("setup_ms", "table_upload_ms", "launch_ms", "kernel_ms", "readback_ms", "closing_ms")


@pytest.fixture()
def eng():
    import montecarlocuda_amd as mc
    e = mc.Engine(0)
    yield e
    e.close()


def _sum(k):
    return sum(k[p] for p in PARTS)


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_parts_add_up_to_the_call(eng, X):
    first = eng.vanilla(VAN, 10 ** 8, precision=X)
    k = eng.last_call_stats()
    assert k["first_call"] == 1 and k["context_create_ms"] > 0.1          # creating the context is reported, apart from the call
    # the first launch of a kernel loads its code object inside the launch call: counted as launch, not (again) as kernel
    assert k["wall_ms"] == pytest.approx(first.wall_ms, rel=1e-6) and k["kernel_ms"] <= first.kernel_ms * (1 + 1e-6)
    assert _sum(k) == pytest.approx(k["wall_ms"], rel=0.05), k
    for _ in range(3):
        e = eng.vanilla(VAN, 10 ** 8, precision=X)
        k = eng.last_call_stats()
        assert k["first_call"] == 0 and k["setup_ms"] == 0 and k["table_upload_ms"] == 0
        assert _sum(k) == pytest.approx(k["wall_ms"], rel=0.05), k
        assert k["kernel_ms"] > 0.3 * k["wall_ms"] and k["kernel_ms"] == pytest.approx(e.kernel_ms, rel=0.02)      # most of a 1e8-path call is its kernel (65 of ~95 us in fp32 with events on)
        assert k["launch_ms"] < 0.3 and k["closing_ms"] < 0.05      # ~15 us and < 1 us on a quiet box; loose for a shared one
    # timing off: no events -- the kernel's time is inside the wait (read-back), the parts still add up
    eng.set_timing(False)
    eng.vanilla(VAN, 10 ** 8, precision=X)
    k = eng.last_call_stats()
    assert k["kernel_ms"] == 0 and k["readback_ms"] > 0.4 * k["wall_ms"] and _sum(k) == pytest.approx(k["wall_ms"], rel=0.05)


def test_cva_table_is_reported_once_per_input(eng):
    eng.cva(CVA, 10 ** 6)
    k1 = eng.last_call_stats()
    eng.cva(CVA, 10 ** 6)
    k2 = eng.last_call_stats()
    assert k1["table_upload_ms"] > 0 and k2["table_upload_ms"] == 0      # the per-date table is built and uploaded when the inputs change
    eng.cva(dict(CVA, v=0.25), 10 ** 6)
    k3 = eng.last_call_stats()
    assert k3["table_upload_ms"] > 0
    for k in (k1, k2, k3):
        assert _sum(k) == pytest.approx(k["wall_ms"], rel=0.05), k


@pytest.mark.parametrize("X", ["f32", "f64"])
def test_first_launch_geometry_call_reports_its_setup(eng, X):
    """The reference's own launch (512 x 128): the first call of a geometry computes the XORWOW jump matrices (once per process) and
    one start state per thread -- the reference's randomSetup, paid there on every call (dp/MonteCarloKernel.cu:285-290,317-323).
    Reported as set-up, waited for before the pricing kernel, so kernel_ms is the pricing kernel's in the first call too."""
    a = eng.run_grid("vanilla", VAN, 512, 128, 195312, X)
    k1 = eng.last_call_stats()
    b = eng.run_grid("vanilla", VAN, 512, 128, 195312, X)
    k2 = eng.last_call_stats()
    assert a.expected == b.expected and a.confidence == b.confidence
    assert k1["setup_ms"] > 0.2 and k2["setup_ms"] == 0, (k1, k2)
    assert _sum(k1) == pytest.approx(k1["wall_ms"], rel=0.05) and _sum(k2) == pytest.approx(k2["wall_ms"], rel=0.05), (k1, k2)
    # the set-up is not inside the kernel stage (it used to sit between the two events)
    assert k1["kernel_ms"] < 1.5 * k2["kernel_ms"] + 0.05, (k1, k2)
    # another geometry: a new set of states (no jump matrices this time)
    eng.run_grid("vanilla", VAN, 256, 64, 1000, X)
    k3 = eng.last_call_stats()
    assert 0 < k3["setup_ms"] < k1["setup_ms"] + 1.0
    # more geometries than the cache holds: the eviction path (erase before free) keeps working, results repeat
    for nb, nt in ((128, 64), (64, 128), (32, 256), (16, 64), (512, 128)):
        r = eng.run_grid("vanilla", VAN, nb, nt, 2000, X)
        assert r.n == nb * 2000
    assert eng.run_grid("vanilla", VAN, 512, 128, 195312, X).expected == a.expected


def test_describe_names_the_resolved_configuration(eng):
    text = eng.describe()
    for word in ("mc_context config:", "device=0", "blocks=2048", "finish=fused", "f64_normals=native", "rng=philox", "basket_static_max=f32:12,f64:8",
                 "basket_tiled_min=9", "created_in_ms="):
        assert word in text, (word, text)
    eng.set_normals("f32")
    eng.set_generator("xorwow")
    assert "f64_normals=f32" in eng.describe() and "rng=xorwow" in eng.describe()


@pytest.mark.parametrize("X", ["f64", "f32"])
def test_legacy_symbols_print_the_stages_under_MC_VERBOSE(X):
    exe = os.path.join(ROOT, "drivers", f"vanillaOpt_{X}")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "drivers")], stdout=subprocess.DEVNULL)
    quiet = subprocess.run([exe, "8", "--no-cpu"], capture_output=True, text=True, timeout=600)
    assert quiet.returncode == 0 and "stages (ms)" not in quiet.stdout and "config" not in quiet.stderr
    one = subprocess.run([exe, "8", "--no-cpu"], capture_output=True, text=True, timeout=600, env=dict(os.environ, MC_VERBOSE="1"))
    assert one.returncode == 0, one.stdout + one.stderr
    lines = [l for l in one.stdout.splitlines() if "stages (ms)" in l]
    assert lines and "set-up" in lines[0] and "read-back" in lines[0] and "= call" in lines[0]
    assert "first call of this context: creating it" in lines[0] and all("first call" not in l for l in lines[1:])
    assert "config" not in one.stderr
    two = subprocess.run([exe, "8", "--no-cpu"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, MC_VERBOSE="2", MC_SEED="0x1234", MC_ANTITHETIC="1"))
    assert two.returncode == 0, two.stdout + two.stderr
    assert "legacy symbols config" in two.stderr and "MC_SEED=0x1234" in two.stderr and "MC_ANTITHETIC=1" in two.stderr
    assert "mc_context config:" in two.stderr and "antithetic=" in two.stderr and "stages (ms)" in two.stdout
    # several devices: the handle's configuration too
    three = subprocess.run([exe, "8", "--no-cpu"], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, MC_VERBOSE="2", MC_DEVICES="0,0", MC_MULTI_REDUCE="host"))
    assert three.returncode == 0, three.stdout + three.stderr
    assert "mc_multi config: devices=[0,0] reduce=host" in three.stderr and "linger_us=5000" in three.stderr


def test_staged_launch_geometry_form_counts_its_normals_pass(eng):
    """The staged form of the launch-geometry mode (normals through HBM, then the engine's kernels; kept as the checker and for shapes
    without a fused kernel) does part of its work before the pricing launch: the call's wall time starts at ITS entry, the parts
    still add up, and what per-path dumps or other calls did before is not charged to it."""
    eng.set_grid_form("staged")
    eng.vanilla_paths(VAN, 1000, precision="f64")          # a call of another kind first (allocates the dump buffer)
    a = eng.run_grid("vanilla", VAN, 64, 128, 4000, "f64")
    k1 = eng.last_call_stats()
    b = eng.run_grid("vanilla", VAN, 64, 128, 4000, "f64")
    k2 = eng.last_call_stats()
    assert a.expected == b.expected and k1["setup_ms"] > 0.2 and k2["setup_ms"] == 0
    for k in (k1, k2):
        assert _sum(k) == pytest.approx(k["wall_ms"], rel=0.05), k
    assert k2["wall_ms"] == pytest.approx(b.wall_ms, rel=1e-6) and k2["launch_ms"] > k2["kernel_ms"] * 0.05      # the normals pass is host-visible time
    eng.set_grid_form("auto")
