"""CPU: the C-ABI library loads, exports every symbol include/anofox_fcst_hip.h declares, keeps the
reference's struct layout and -- without a GPU -- fails loudly instead of
falling back to any CPU path.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol(hiplib):
    L = hiplib.load()
    header = open(os.path.join(ROOT, "include", "anofox_fcst_hip.h")).read()
    declared = set(re.findall(r"\b(anofox_[a-z_0-9]+)\s*\(", header))
    assert declared == set(hiplib.EXPORTED_SYMBOLS), declared ^ set(hiplib.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(L, sym), sym
    assert L.anofox_fcst_version().startswith(b"0.1.0")


def test_struct_layout_matches_reference_header():
    """Offsets measured against the reference's cbindgen header."""
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "anofox_fcst_hip.h"
int main(void) {
  printf("%zu %zu %zu\n", sizeof(ForecastOptions), sizeof(ForecastResult), sizeof(AnofoxError));
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", offsetof(ForecastOptions, model), offsetof(ForecastOptions, ets_model),
    offsetof(ForecastOptions, horizon), offsetof(ForecastOptions, confidence_level), offsetof(ForecastOptions, seasonal_period),
    offsetof(ForecastOptions, auto_detect_seasonality), offsetof(ForecastOptions, include_fitted), offsetof(ForecastOptions, include_residuals),
    offsetof(ForecastOptions, window), offsetof(ForecastOptions, seasonal_periods_str), offsetof(ForecastOptions, model_pool),
    offsetof(ForecastOptions, laplace_variant));
  printf("%zu %zu %zu %zu %zu %zu\n", offsetof(ForecastResult, n_forecasts), offsetof(ForecastResult, n_fitted), offsetof(ForecastResult, model_name),
    offsetof(ForecastResult, aic), offsetof(ForecastResult, bic), offsetof(ForecastResult, mse));
  printf("%d %d %d %d\n", (int)INVALID_INPUT, (int)COMPUTATION_ERROR, (int)INVALID_MODEL, (int)INTERNAL_ERROR);
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split("\n")
    assert out[0] == "184 144 260"
    assert out[1] == "0 32 40 48 56 60 61 62 64 68 132 164"
    assert out[2] == "40 48 56 120 128 136"
    assert out[3] == "2 3 5 10"


def test_additive_structs_match_their_ctypes_mirrors(hiplib):
    """The backend's own structs (AnofoxHipStats -- round 6: `reserved` became `y_storage`, same place --, AnofoxHipLaneStats,
    AnofoxHipInspection): size and the offsets that matter, as gcc lays out include/anofox_fcst_hip.h against lib.py's ctypes mirrors."""
    src = r"""
#include <stdio.h>
#include <stddef.h>
#include "anofox_fcst_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(AnofoxHipStats), offsetof(AnofoxHipStats, fit_kernel_launches), offsetof(AnofoxHipStats, y_storage),
         offsetof(AnofoxHipStats, total_iters), offsetof(AnofoxHipStats, min_pass_bytes));
  printf("%zu %zu\n", sizeof(AnofoxHipLaneStats), sizeof(AnofoxHipInspection));
  return 0; }"""
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split("\n")
    S = hiplib.AnofoxHipStats
    assert out[0] == f"{C.sizeof(S)} {S.fit_kernel_launches.offset} {S.y_storage.offset} {S.total_iters.offset} {S.min_pass_bytes.offset}"
    assert out[1] == f"{C.sizeof(hiplib.AnofoxHipLaneStats)} {C.sizeof(hiplib.AnofoxHipInspection)}"


def test_no_gpu_fails_loudly(hiplib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from anofox_forecast_amd import api
    r = api.forecast_series([1.0, 2.0, 3.0, 4.0], hiplib.make_options("Naive", 2))
    assert not r["ok"] and r["code"] == hiplib.INTERNAL_ERROR and "no CPU fallback" in r["message"]
    # argument errors are still reported first, like the reference
    r = api.forecast_series([1.0, 2.0, 3.0, 4.0], hiplib.make_options("NoSuchModel", 2))
    assert r["code"] == hiplib.INVALID_MODEL and "Unknown model" in r["message"]
    r = api.forecast_series([1.0, 2.0], hiplib.make_options("Naive", 2))
    assert r["code"] == hiplib.INSUFFICIENT_DATA
    err = hiplib.AnofoxError()
    ok = hiplib.load().anofox_ts_forecast(None, None, 0, None, None, C.byref(err))
    assert not ok and err.code == hiplib.NULL_POINTER


def test_shard_ranges_of_the_multi_device_batch_entry(hiplib):
    """anofox_hip_shard_range is the rule the batch entry cuts a batch by (contiguous
    [g ceil(N / G), (g + 1) ceil(N / G)) clipped to N) -- the same rule as dist.shard_range: the ranges tile [0, N) in order,
    none is larger than ceil(N / G), trailing shards may be empty.  Host logic only: no GPU needed."""
    from anofox_forecast_amd import dist
    for n in (0, 1, 7, 64, 1000, 30490, 1000003):
        for g in (1, 2, 3, 4, 8, 16):
            per = -(-n // g) if g else n
            pos = 0
            for k in range(g):
                lo, hi = hiplib.shard_range(n, g, k)
                assert lo == min(n, k * per) and hi == min(n, lo + per) and lo == pos, (n, g, k, lo, hi)
                assert (lo, hi) == dist.shard_range(n, k, g)
                pos = hi
            assert pos == n


def test_device_list_api_without_a_gpu(hiplib):
    """The device list is validated against the visible devices: without a GPU no ordinal is acceptable, the empty list (the
    default: the caller's current device) always is; release_caches and the ARIMA method setter are safe to call on an idle library."""
    import torch
    L = hiplib.load()
    if not torch.cuda.is_available():
        assert not L.anofox_hip_set_devices((C.c_int * 1)(0), 1)
    assert L.anofox_hip_set_devices(None, 0) and L.anofox_hip_get_devices(None, 0) == 0
    L.anofox_hip_release_caches()
    assert L.anofox_hip_set_default_arima_method(1) and L.anofox_hip_set_default_arima_method(0) and not L.anofox_hip_set_default_arima_method(2)
    err = hiplib.AnofoxError()
    assert not L.anofox_hip_batch_set_arima_method(None, 0, C.byref(err)) and err.code == hiplib.NULL_POINTER
    assert not L.anofox_hip_batch_run_many(None, 1, None)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may import, link or call it."""
    pkg = os.path.join(ROOT, "anofox-forecast_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, os.path.join(dp, f)
                # no source under the package reaches into oracle/: neither an #include nor a path (comments that NAME an
                # oracle file for provenance, "oracle/ets.c", are the only mentions allowed)
                assert '#include "../../oracle' not in txt and '#include "oracle' not in txt, os.path.join(dp, f)
                for line in txt.splitlines():
                    if "oracle/" not in line:
                        continue
                    code = line.split("//")[0].split("#")[0] if f.endswith((".hip", ".hpp", ".cpp", ".h", ".py")) or f == "Makefile" else line
                    assert "oracle/" not in code, (os.path.join(dp, f), line)
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libanofox_fcst_hip.so")]).decode() if os.path.exists(os.path.join(pkg, "libanofox_fcst_hip.so")) else ""
    assert "liboracle" not in out


def build_c_caller(out_dir):
    """Compile tests/c_abi/caller.c against include/ and the in-tree shared library; returns the binary path."""
    exe = os.path.join(out_dir, "caller")
    pkg = os.path.join(ROOT, "anofox-forecast_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "caller.c"), "-L", pkg, "-lanofox_fcst_hip",
                           "-Wl,-rpath," + pkg, "-o", exe])
    return exe


def test_plain_c_caller_links_and_reports_errors(hiplib):
    """A C program written like the reference's binding links against the library with nothing but the
    header; without a GPU it gets INTERNAL_ERROR (no fallback), and argument errors still come first."""
    import torch
    hiplib.load()
    with tempfile.TemporaryDirectory() as d:
        exe = build_c_caller(d)
        out = subprocess.run([exe, "NoSuchModel"], capture_output=True, text=True, timeout=120).stdout
        assert out.startswith("ERR %d " % hiplib.INVALID_MODEL) and "Unknown model" in out, out
        if not torch.cuda.is_available():
            out = subprocess.run([exe, "Naive", "0"], capture_output=True, text=True, timeout=120).stdout
            assert out.startswith("ERR %d" % hiplib.INTERNAL_ERROR) and "no CPU fallback" in out, out


def test_ingest_under_address_and_ub_sanitizers():
    """csrc/ingest.hip is host-only C++: build it with g++ ASan+UBSan together with tests/c_abi/ingest_san.cpp, which
    replays random chunked appends against the reference's collection rule (ts_forecast_native.cpp:502-600)."""
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "ingest_san")
        subprocess.check_call(["g++", "-std=c++17", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                               "-x", "c++", os.path.join(ROOT, "anofox-forecast_amd", "csrc", "ingest.hip"),
                               "-x", "c++", os.path.join(ROOT, "tests", "c_abi", "ingest_san.cpp"), "-o", exe])
        r = subprocess.run([exe, "150"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.startswith("OK 150 rounds"), (r.stdout, r.stderr[-2000:])


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_device_keyed_caches_on_four_fake_devices(sanitizer):
    """csrc/host_resources.hpp -- DeviceGuard, the caching device allocator, the pinned cache, the stream sets, the device list --
    compiled against tests/c_abi/fake_hip.h (a host-only stand-in for the ~20 HIP calls it makes: four "devices" with 64 MB
    each) and driven by eight host threads under ThreadSanitizer, then under ASan + UBSan (tests/c_abi/resources_mt.cpp): a cached
    block only goes back to its own device, caps / LRU eviction / the bounded deferral hold per device, a second free never
    reaches hipFree, out-of-memory empties the cache and retries, one priority stream set per device, nothing leaks.  The real
    runtime only ever showed this code ONE device.  Test infrastructure: the product never sees the fake."""
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "resources_mt")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-sanitize-recover=all", "-pthread",
                               os.path.join(ROOT, "tests", "c_abi", "resources_mt.cpp"), "-o", exe])
        env = dict(os.environ, FAKE_HIP_DEVICES="4", FAKE_HIP_DEVICE_MB="64", TSAN_OPTIONS="halt_on_error=1")
        env.pop("ANOFOX_HIP_CACHE_GB", None)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and r.stdout.startswith("resources_mt: ok"), (r.stdout, r.stderr[-3000:])


def test_duckdb_binding_parses():
    """binding/*.cpp have never been through the extension's build (no DuckDB headers in this image).  This is the
    next best thing: `g++ -fsyntax-only` against the REFERENCE's own helper headers (src/include/ts_forecast_native.hpp,
    ts_fill_gaps_native.hpp, anofox_fcst_ffi.h -- read where they lie, in this container only) and a declaration-only stand-in for
    DuckDB's API (tests/c_abi/duckdb_stub/, test infrastructure, written from DuckDB's public interface).  It proves the file is
    well-formed C++, that every DuckDB call it makes exists with a compatible shape in the stand-in, and -- the part that is NOT a
    stand-in -- that include/anofox_fcst_hip.h coexists with the reference's anofox_fcst_ffi.h in one translation unit (the shared
    include guard: block 1 steps aside)."""
    import shutil
    import subprocess
    ref = "/root/reference/src/include"
    if not os.path.isdir(ref):
        pytest.skip("the reference tree is not on this machine (GPU box): its helper headers are read in place, never copied")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    # round 6: the scalar of route A (one batch per DataChunk) and the macro that sends ts_forecast_by to the batch route
    for unit in ("ts_forecast_native_hip.cpp", "ts_forecast_scalar_hip.cpp", "ts_macros_hip.cpp"):
        cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "tests", "c_abi", "duckdb_stub"), "-I", ref,
               "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "binding", unit)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (unit, out.stderr[-3000:])
