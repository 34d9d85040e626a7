"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/splash_hip.h
declares; without a GPU every compute entry point refuses loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    h = open(os.path.join(ROOT, "include", "splash_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(splash_[a-z_A-Z0-9]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    import rsplash_amd
    rsplash_amd.build_library()
    from rsplash_amd import _lib
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in include/splash_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names
    assert _lib.lib.splash_hip_abi_version() == 6


def test_product_never_touches_the_oracle():
    """the shipped package must not import, link or execute anything under oracle/"""
    pkg = os.path.join(ROOT, "rsplash_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.lower().replace("# oracle", ""), os.path.join(dp, f)


def test_no_gpu_means_loud_failure(bourne):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rsplash_amd import api
    assert api.device_count() == 0
    b, s = bourne, bourne["site"]
    with pytest.raises(api.SplashError, match="no HIP device"):
        api.splash_point(b["sw_in"], b["Ta"], b["P"], s["latitude"], s["elev_m"], soil_data=b["soil"],
                         time_index=b["dates"])
    with pytest.raises(api.SplashError, match="no HIP device"):
        api.SPLASH.moist_surf(1.0, 10.0, -800.0, 150.0, 250.0, 20.0, 0.37)
    with pytest.raises(api.SplashError):
        api.Calendar(b["doy"], b["year"], b["month"])
    # the two halves of the grid call (ABI 3) refuse at submit time, before any thread is started
    n, nt = 4, len(b["dates"])
    tile = lambda x: np.repeat(np.asarray(x, dtype=np.float64)[:, None], n, axis=1)
    with pytest.raises(api.SplashError, match="no HIP device"):
        api.splash_grid(tile(b["sw_in"]), tile(b["Ta"]), tile(b["P"]), np.full(n, s["elev_m"]),
                        np.repeat(np.asarray(b["soil"], dtype=np.float64)[:, None], n, axis=1),
                        np.full(n, s["latitude"]), np.zeros(n), np.zeros(n), time_index=b["dates"], submit=True)


def test_time_fields_and_month_ids(bourne):
    from rsplash_amd import api
    doy, year, month = api.time_fields(bourne["dates"])
    assert np.array_equal(doy, bourne["doy"]) and np.array_equal(year, bourne["year"])
    assert np.array_equal(month, bourne["month"])
    assert np.array_equal(api.month_ids(year, month), bourne["month_id"])


def test_host_orbit_table_matches_oracle(oracle):
    """the per-day orbit terms the calendar tabulates on the host (src/SOLAR.cpp:98-124) against the
    oracle's (which is bit-identical to the real SOLAR class): dr bitwise, sin/cos(delta) to 1 ulp"""
    import math
    from rsplash_amd import api
    pir = math.pi / 180.0
    for year in (0, 1999, 2000, 2001, 2004, 2100):
        for doy in (1, 59, 60, 80, 172, 266, 355, 365, 366):
            dr, sd, cd = api.orbit_day(doy, year)
            kN, nu, lam, dr_o, delta = oracle.orbit(doy, year)
            assert dr == dr_o, (doy, year)
            assert abs(sd - math.sin(delta * pir)) <= 2.3e-16 and abs(cd - math.cos(delta * pir)) <= 2.3e-16


def build_cpp_mirror(outdir):
    """g++ the C++ host-mirror driver against the built library; returns the executable path"""
    import subprocess
    import rsplash_amd
    so = rsplash_amd.build_library()
    d = os.path.dirname(so)
    exe = os.path.join(outdir, "cpp_mirror")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests/csrc/test_cpp_mirror.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + d, "-lrsplash_hip",
                           "-Wl,-rpath," + d, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe])
    return exe


def test_cpp_host_mirror_compiles_and_links(tmp_path):
    """include/splash.hpp (class SPLASH with the Rcpp module's method signatures) against the .so"""
    import subprocess
    r = subprocess.run([build_cpp_mirror(str(tmp_path))], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr


def test_r_wrapper_labels_results_positionally():
    """r/splash_hip.R (no R interpreter here: checked as text): the list splash.grid() returns is
    labelled with `.splash_outputs$ret`, which must name, position by position, the brick that
    `.splash_outputs$key` -- the order the C shim returns them in -- holds there (round 2 had bflow and
    cond swapped), and the shim's own order must be that of SPLASH_OUT_* in the header."""
    import re
    txt = open(os.path.join(ROOT, "r", "splash_hip.R")).read()
    tab = txt[txt.index(".splash_outputs <- data.frame("):]
    tab = tab[:tab.index("stringsAsFactors")]

    def col(name):
        m = re.search(r"\b" + name + r"\s*=\s*c\(([^)]*)\)", tab)
        return re.findall(r'"([^"]*)"', m.group(1))
    key, ret, stem = col("key"), col("ret"), col("stem")
    assert key == ["wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "sm_lim"]
    assert ret == [{"netr": "netrad"}.get(k, k) for k in key]
    assert stem == [{"wn": "sm", "snow": "swe", "netr": "netrad"}.get(k, k) for k in key]
    # the order of the reference's return value (R/splash.grid.R:449) is restored BY NAME
    assert 'out[c("wn", "ro", "pet", "aet", "snow", "bflow", "cond", "netrad", "sm_lim")]' in txt
    # the pipelined loop: blocks are submitted ahead and collected in order; the shim registers both halves
    assert '.Call("C_splash_grid_submit"' in txt and '.Call("C_splash_grid_wait"' in txt and "inflight" in txt
    shim_txt = open(os.path.join(ROOT, "r", "splash_shim.c")).read()
    assert '{"C_splash_grid_submit", (DL_FUNC)&C_splash_grid_submit, 12}' in shim_txt
    assert '{"C_splash_grid_wait", (DL_FUNC)&C_splash_grid_wait, 1}' in shim_txt
    n_args = len(re.search(r'\.Call\("C_splash_grid_submit",(.*?)\)\n\}', txt, re.S).group(1).split(","))
    assert n_args >= 12
    hdr = open(os.path.join(ROOT, "include", "splash_hip.h")).read()
    m = re.search(r"#define SPLASH_NOUT 9\s*/\*\s*([a-z_, ]+?)\s*\(", hdr)
    assert [x.strip() for x in m.group(1).split(",")] == key
    shim = open(os.path.join(ROOT, "r", "splash_shim.c")).read()
    for m in re.finditer(r'names\[\] = \{([^}]*)\}', shim):
        got = re.findall(r'"([^"]*)"', m.group(1))
        if got[0] == "wn":
            assert got[:9] == key


def test_grid_job_waits_before_it_lets_go_of_the_arrays():
    """A GridJob dropped without wait() -- e.g. the second of two jobs in flight when the first one's
    wait() raised -- must wait for the library's thread BEFORE the arrays it reads and writes are
    released (include/splash_hip.h: every array stays valid until _wait returns).  No
    device here: the wait function is injected and records the order of events."""
    import gc
    import weakref
    from rsplash_amd import api

    class Arr:   # stands for the numpy arrays the job keeps alive
        pass
    events = []

    def make():
        a = Arr()
        weakref.finalize(a, lambda: events.append("arrays released"))

        def wait_fn(job, err, errlen):
            events.append("waited")
            return 0
        return api.GridJob(object(), (a,), lambda: {"ok": True}, wait_fn=wait_fn)

    j = make()
    del j
    gc.collect()
    assert events == ["waited", "arrays released"], events          # dropped: waited first
    del events[:]
    with make() as j:
        pass
    gc.collect()
    assert events[:1] == ["waited"] and events.count("waited") == 1, events   # context manager: once
    del events[:]
    j = make()
    assert j.wait() == {"ok": True}
    with pytest.raises(api.SplashError):
        j.wait()
    j.close()
    del j
    gc.collect()
    assert events == ["waited", "arrays released"], events          # waited once, not again by close/del

    def failing(job, err, errlen):
        events.append("waited")
        return -3
    del events[:]
    j = api.GridJob(object(), (Arr(),), lambda: {}, wait_fn=failing)
    with pytest.raises(api.SplashError):
        j.wait()
    del j
    gc.collect()
    assert events == ["waited"], events


def test_r_shim_syntax_and_registration():
    """r/splash_shim.c has never met R's headers in this image (no R).  It is put through
    `gcc -fsyntax-only -Wall -Wextra` against DECLARATIONS-ONLY headers of our own
    (tests/csrc/r_decls: the ~30 functions of R's public C API the shim uses, with their documented
    signatures) and the real include/splash_hip.h, so that a typo, a wrong arity or a wrong argument
    type in the submit/_wait/finalizer code cannot ship; and the arities in its registration table must
    be those of the definitions (R checks them at .Call time: src/RcppExports.cpp:9-25 is what the
    table replaces).  Syntax-checked, not run."""
    import subprocess
    shim = os.path.join(ROOT, "r", "splash_shim.c")
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type",
                        "-I" + os.path.join(ROOT, "tests", "csrc", "r_decls"), "-I" + os.path.join(ROOT, "include"), shim],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    txt = re.sub(r"/\*.*?\*/", "", open(shim).read(), flags=re.S)
    defs = {m.group(1): m.group(2).count("SEXP") for m in re.finditer(r"^SEXP (C_\w+)\(([^)]*)\)\s*\{", txt, re.M)}
    table = {m.group(1): int(m.group(2)) for m in re.finditer(r'\{"(C_\w+)", \(DL_FUNC\)&\1, (\d+)\}', txt)}
    assert table and table == defs, (table, defs)
    # every entry point the R wrapper calls is registered
    wrapper = open(os.path.join(ROOT, "r", "splash_hip.R")).read()
    for name in set(re.findall(r'\.Call\("(C_\w+)"', wrapper)):
        assert name in table, name
    # and the negative control: a wrong arity is caught
    bad = txt.replace("R_ClearExternalPtr(ptr);", "R_ClearExternalPtr(ptr, 1);")
    assert bad != txt
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", "-I" + os.path.join(ROOT, "tests", "csrc", "r_decls"),
                        "-I" + os.path.join(ROOT, "include"), "-"], input=bad, capture_output=True, text=True)
    assert r.returncode != 0


def test_r_shim_runs_against_a_mock_r_api(oracle, bourne, tmp_path):
    """r/splash_shim.c COMPILED AND EXECUTED (no R in this image: against tests/csrc/r_mock/r_mock.c, a mock of the ~30
    functions of R's C API the shim uses, and the declarations-only headers of tests/csrc/r_decls/).  The mock's driver
    calls the shim's .Call entry points through the shim's own registration table, with the host backend (devices = -1L /
    backend = 1L: no GPU needed), and checks on its side: registration arities, PROTECT balance of every call, result list
    names and matrix dimensions, submit/wait == run bit for bit, the finalizer of a dropped handle, and that bad arguments,
    a second collect, and the default backend without a device are R errors.  Here: what came back through the SEXPs equals
    what the Python mirror gets from the same C ABI -- splash.point on Bourne (R/splash.point.R:29), a row block
    (R/splash.grid.R:277-308), and the series in two time chunks (ABI 5) == the undivided block, bit for bit.
    What this cannot show is that real R behaves like the mock."""
    import subprocess
    import rsplash_amd
    from rsplash_amd import api
    so = rsplash_amd.build_library()
    d = os.path.dirname(so)
    exe = str(tmp_path / "r_mock")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wno-cast-function-type", "-Werror",
                           "-I" + os.path.join(ROOT, "tests/csrc/r_decls"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "r/splash_shim.c"), os.path.join(ROOT, "tests/csrc/r_mock/r_mock.c"),
                           "-L" + d, "-lrsplash_hip", "-Wl,-rpath," + d, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                           "-o", exe])
    b, s = bourne, bourne["site"]
    nt = 800
    cut = 397                                     # 2009-02-01 (2008 is a leap year): the second chunk starts on the first day of a month
    assert b["dates"][cut].astype("datetime64[D]").astype(object).day == 1
    H, W = 2, 8
    n = H * W
    doy, year, month = b["doy"][:nt], b["year"][:nt], b["month"][:nt]
    st = oracle.synth_statics(1234, H, W, 0, n, 1)
    sw, tc, pn = oracle.synth_forcing(1234, H, W, 0, n, 0, doy, st["lat"], st["elev"])
    # the whole series' snowfall threshold per cell (R/splash.point.R:120-122), from the host backend's own aux
    dates = b["dates"][:nt]
    full = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"],
                           time_index=dates, sim_control=dict(monthly_out=False), want_aux=True, backend="host")
    tt = full["aux"]["Tt"]
    site = [s["latitude"], s["elev_m"], s["slop_250m"], s["asp_250m"], s["resolution"], s["Aups_250m"]]
    soil1 = np.asarray(b["soil"], dtype=np.float64)
    stat = np.stack([st["lat"], st["elev"], st["slop"], st["asp"], st["resolution"]])
    blob = np.concatenate([[nt, n, cut], site, soil1, doy, year, month, b["sw_in"][:nt], b["Ta"][:nt], b["P"][:nt],
                           stat.ravel(), st["soil"].ravel(), st["au"].ravel(), tt, sw.ravel(), tc.ravel(), pn.ravel()])
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    blob.astype("f8").tofile(fin)
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "0 problems" in r.stdout
    got = np.fromfile(fout)
    names = api.OUT_NAMES
    pt = got[:9 * nt].reshape(9, nt)
    ref = api.splash_point(b["sw_in"][:nt], b["Ta"][:nt], b["P"][:nt], s["latitude"], s["elev_m"], s["slop_250m"],
                           s["asp_250m"], soil1, s["Aups_250m"], s["resolution"], time_index=dates, backend="host")
    for i, k in enumerate(names):
        assert np.array_equal(pt[i], ref[k], equal_nan=True), k
    at = 9 * nt
    gr = got[at:at + 9 * nt * n].reshape(9, nt, n)
    at += 9 * nt * n
    ch = got[at:at + 9 * nt * n].reshape(9, nt, n)       # (each series: chunk 1 then chunk 2)
    at += 9 * nt * n
    se = got[at:at + 5 * n].reshape(5, n)
    for i, k in enumerate(names):
        assert np.array_equal(gr[i], full[k], equal_nan=True), k
        assert np.array_equal(ch[i], full[k], equal_nan=True), "time chunks: " + k
    assert np.array_equal(se, full["state_end"], equal_nan=True)
    at += 5 * n
    un = got[at:at + 3 * nt].reshape(3, nt)                # unSWC of the site's wn to 0.49 m, host backend
    at += 3 * nt
    ug = got[at:at + 4 * nt * n].reshape(4, nt, n)           # unSWC.grid of the block's wn brick to 0.3 m, host backend
    at += 4 * nt * n
    assert at == got.size
    ugref = api.unSWC_grid(st["soil"], 0.3, full["wn"], backend="host")
    for i, k in enumerate(("w_z", "wtd", "Se", "theta_mean")):
        assert np.array_equal(ug[i], ugref[k], equal_nan=True), "unSWC.grid: " + k
    uref = api.unSWC(soil1, 0.49, ref["wn"], backend="host")
    for i, k in enumerate(("wn_z", "wtd", "theta_i")):
        assert np.array_equal(un[i], uref[k], equal_nan=True), k
