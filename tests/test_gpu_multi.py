"""libmc_multi.so (include/mc_multi.h): one pricing call over several GPUs from ONE C process, closed by one
RCCL all-reduce -- exercised from C (tests/c/multi_check.c, no Python or torch in that process), through the
legacy symbols (MC_DEVICES) and through the strong-scaling driver.  On a one-GPU box the RCCL path runs with a
communicator of one and the G > 1 sharding with a repeated device + host reduction; with more devices visible
the same binary also runs all of them under RCCL and asserts multi == single (1e-12 rel in fp64)."""
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "montecarlocuda_amd", "csrc")
pytestmark = pytest.mark.gpu


def build_check(tmp_path):
    exe = tmp_path / "multi_check"
    subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-Wall", f"-I{ROOT}/include", os.path.join(ROOT, "tests", "c", "multi_check.c"),
                           "-o", str(exe), f"-L{CSRC}", "-lmc_multi", "-lmc_mi355x", "-lm", f"-Wl,-rpath,{CSRC}",
                           f"-Wl,-rpath-link,{CSRC}:/opt/rocm/lib"])
    return exe


def test_c_multi_device_path_matches_single_device(tmp_path):
    exe = build_check(tmp_path)
    # linger 2 ms (the default is 100 ms): the program's 5 ms pauses then let the launcher threads park, so that the wake-up
    # path (and the caller taking over a job whose worker is still waking up) is exercised as well as the spinning one
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=dict(os.environ, MC_MULTI_LINGER_US="2000"))
    print(out.stdout[-6000:])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "all checks passed" in out.stdout and "MISMATCH" not in out.stdout
    assert "{0} rccl" in out.stdout and "{0,0,0} host" in out.stdout
    if int(re.search(r"visible devices: (\d+)", out.stdout).group(1)) > 1:
        assert "all rccl" in out.stdout


@pytest.mark.parametrize("ranks", [3, 8])
def test_grouped_allreduce_call_sequence_through_a_test_double(tmp_path, ranks):
    """The G > 1 control flow of run_sharded's collective branch -- G calls of ncclAllReduce inside one group, rank g's own send /
    receive buffers, stream and communicator, the publish of device 0's reduced triple, the cross-check against the host sum --
    executed with THREE and with EIGHT ranks (the node's own size: eight launcher threads, nine pinned slots) on the one GPU.
    Real RCCL refuses a repeated device, so a test double of its six entry points (tests/cpp/rccl_mock.hip: event-ordered sum of the
    ranks' buffers in rank order) is preloaded in front of librccl.so and MC_MULTI_ALLOW_REPEATED_DEVICES=1 lifts the library's own
    duplicate check -- in a TEST BUILD of the library only (`make libmc_multi_testhooks`: -DMC_MULTI_TEST_HOOKS, built here into a
    temporary directory that goes in front on LD_LIBRARY_PATH; the shipped libmc_multi.so has no such switch, see the next test).
    This proves the call sequence and the indexing, NOT RCCL or xGMI: the multi-rank collective itself stays unexercised on
    one-GPU boxes."""
    mock = tmp_path / "librccl_mock.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC",
                           os.path.join(ROOT, "tests", "cpp", "rccl_mock.hip"), "-o", str(mock)])
    exe = build_check(tmp_path)
    hooks = tmp_path / "hooks"
    hooks.mkdir()
    subprocess.check_call(["make", "-C", CSRC, "libmc_multi_testhooks", f"TESTHOOKS_OUT={hooks / 'libmc_multi.so'}"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=str(mock), MC_MULTI_ALLOW_REPEATED_DEVICES="1", RCCL_MOCK_VERBOSE="1", MC_MULTI_LINGER_US="2000",
               MULTI_CHECK_RANKS=str(ranks), LD_LIBRARY_PATH=f"{hooks}:{os.environ.get('LD_LIBRARY_PATH', '')}")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=env)
    print(out.stdout[-6000:])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "all checks passed" in out.stdout and "MISMATCH" not in out.stdout
    rep = "{" + ",".join(["0"] * ranks) + "}"
    assert f"{rep} mock collective" in out.stdout and "grouped all-reduce through the preloaded test double" in out.stdout
    assert f"communicator set of {ranks} ranks" in out.stderr          # the double, not RCCL, served the repeated-device handle
    # one launcher thread per rank (the boxes grant 16 CPUs) -- or none at all, by design, on a grant of fewer CPUs than ranks + 1
    assert f"launcher threads: {rep} default {ranks}," in out.stdout or (f"launcher threads: {rep} default 0," in out.stdout and "fewer CPUs" in out.stdout)
    groups = re.findall(r"rccl_mock: (\d+) grouped all-reduces of (\d+) calls in all", out.stderr)
    # hundreds of grouped calls, and more all-reduce calls than groups: groups of `ranks` went through (the {0} handle's are of one)
    assert groups and max(int(g) for g, _ in groups) > 100 and max(int(c) - int(g) for g, c in groups) > 100 * (ranks - 1), out.stderr[-2000:]


def test_shipped_library_has_no_repeated_devices_switch(tmp_path):
    """MC_MULTI_ALLOW_REPEATED_DEVICES is a switch of the test build only: the shipped libmc_multi.so ignores it, so {0,0,0} under
    RCCL is refused by the library's own duplicate check (a clear MC_ERR_INVALID, not an opaque ncclCommInitAll failure)."""
    assert b"MC_MULTI_ALLOW_REPEATED_DEVICES" not in open(os.path.join(CSRC, "libmc_multi.so"), "rb").read()
    exe = build_check(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=dict(os.environ, MC_MULTI_ALLOW_REPEATED_DEVICES="1"))
    assert out.returncode != 0 and "listed twice" in out.stdout + out.stderr, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("X", ["f64", "f32"])
def test_legacy_symbols_shard_over_MC_DEVICES(X):
    """dev_vanillaOpt / dev_basketOpt / dev_cvaEquityOption with MC_DEVICES set go through libmc_multi.so (loaded
    on demand); the C drivers' GPU legs must print the single-device numbers."""
    exe = {p: os.path.join(ROOT, "drivers", f"{p}Opt_{X}") for p in ("vanilla", "basket", "cva")}
    if not all(os.path.exists(e) for e in exe.values()):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "drivers")], stdout=subprocess.DEVNULL)

    def gpu_numbers(prog, env):
        out = subprocess.run([prog, "8", "--no-cpu"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        return [line for line in out.stdout.splitlines() if re.match(r"^-?\d+\.\d+", line)][:2], out.stdout
    for prog in exe.values():
        single, _ = gpu_numbers(prog, {})
        multi, text = gpu_numbers(prog, {"MC_DEVICES": "0", "MC_VERBOSE": "1"})
        assert single and single == multi, (prog, single, multi)
        three, _ = gpu_numbers(prog, {"MC_DEVICES": "0,0,0", "MC_MULTI_REDUCE": "host"})
        assert [round(float(a), 4) for a in three] == [round(float(a), 4) for a in single]
    bad = subprocess.run([exe["vanilla"], "8", "--no-cpu"], capture_output=True, text=True, env=dict(os.environ, MC_DEVICES="0,99"))
    assert bad.returncode == 1 and "out of range" in bad.stderr


def test_strong_scaling_driver_emits_json():
    exe = os.path.join(ROOT, "drivers", "multiBench")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "drivers")], stdout=subprocess.DEVNULL)
    out = subprocess.run([exe, "--small", "--reps", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [json.loads(line) for line in out.stdout.splitlines() if line.startswith("{")]   # RCCL prints a version banner
    shards = [r for r in rows if "shard_of" in r]     # what one device does at G = 2, 4, 8, measured on the first device alone
    assert {(r["shard_of"], r["workload"][:2]) for r in shards} == {(g, w) for g in (2, 4, 8) for w in ("C4", "C5")}
    assert all(r["wall_ms_median"] > 0 and 0.05 < r["device_side_efficiency"] < 1.2 for r in shards)
    runs = [r for r in rows if "workload" in r and "shard_of" not in r]
    assert len(runs) >= 4 and all(r["wall_ms_median"] > 0 and r["rccl_vs_host_rel"] <= 1e-12 for r in runs)
    # the collective as the calling thread saw it (last device's own triple -> all-reduced triple): RCCL over one rank + the publish kernel here
    assert all(0 < r["collective_us"] < 500 and r["device_delivery_us"][0] > 0 for r in runs), [r.get("collective_us") for r in runs]
    assert any(r["workload"].startswith("C4 basket n=16") and 9.5 < r["value"] < 9.9 for r in runs)
    assert any(r["workload"].startswith("C5 CVA") and 0.18 < r["value"] < 0.20 for r in runs)


def test_c_library_over_all_devices_real_rccl_with_fanout_trace(tmp_path):
    """multi_check's "all rccl" leg on a box with >= 2 devices: ncclCommInitAll over every visible device, the grouped all-reduce of
    the triples over xGMI, multi == single to 1e-12 -- with MC_MULTI_TRACE=1, so that the log keeps every call's fan-out (when each
    launcher thread saw the call, when its launch was enqueued).  Skips on one-GPU boxes; there the G > 1 collective stays
    unexercised."""
    import ctypes
    n = ctypes.CDLL(os.path.join(CSRC, "libmc_mi355x.so")).mc_device_count()
    if n < 2:
        pytest.skip(f"needs >= 2 visible devices for a communicator of more than one rank ({n} visible)")
    exe = build_check(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=1200, env=dict(os.environ, MC_MULTI_TRACE="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    print(out.stdout[-6000:])
    print(out.stderr[-6000:])          # the fan-out trace (one line per call)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "all checks passed" in out.stdout and "MISMATCH" not in out.stdout and "all rccl" in out.stdout
    assert f"visible devices: {n}" in out.stdout and "mc_multi fan-out" in out.stderr
