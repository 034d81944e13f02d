"""Test session plumbing.

GPU tests run in CHILD processes, one per test file (and one per test for the tests marked `own_process`: the in-process
200 M / 1 B-record jobs): `pytest -m gpu` in the parent collects, orders and reports as usual, but the parent never touches
the GPU — each unit's tests run in a fresh `python -m pytest <node ids>` child that the parent starts (an ordinary child
process, never an exec of a process that has initialised the GPU) and whose per-test reports it replays.  A GPU fault, a
hang or a leaked pin in one unit ends that unit's child; the other files' results are still there (round 5's single
process lost 69 tests behind one fault).  The files run in the order of SURVEY section 8: the 8a parity tests first.

FASTF_TEST_INPROCESS=1: everything in this process, as before (debugging one test under a tool).
"""
import json
import os
import subprocess
import sys
import tempfile
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The HIP runtime pins pageable host memory of GPU_PINNED_MIN_XFER_SIZE (MiB) or more on the fly for a copy and lets the GPU
# write the caller's heap pages; with the threshold out of reach every pageable copy (torch's own included) is staged through
# the runtime's pinned buffers instead.  The repo's code does not issue such copies (fastf_amd/hostmem.py); this covers the
# ones inside torch that the tests do not control.  Read when the runtime initialises: set before anything touches HIP.
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1000000")
# the library's pin ledger aborts on a violation and the *_pinned / lend / gather entries check their pointers (umi_engine.hip)
os.environ.setdefault("FASTF_DEBUG_PINS", "1")

CHILD = os.environ.get("FASTF_TEST_CHILD") == "1"
INPROCESS = os.environ.get("FASTF_TEST_INPROCESS") == "1"
UNIT_TIMEOUT_S = int(os.environ.get("FASTF_TEST_UNIT_TIMEOUT", "1500"))

# SURVEY section 8 order: the record loop / kernels first, then the front end, the other commands, multi-GPU, end to end
ORDER = ["test_gpu_parity", "test_gpu_kernels", "test_gpu_records", "test_gpu_tags", "test_gpu_inflate", "test_gpu_dist",
         "test_gpu_multi", "test_gpu_e2e"]


def pytest_addoption(parser):
    parser.addoption("--fastf-report", default=None, help="(child runs) write one JSON line per test phase to this file")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "own_process: (gpu tests) run this test in a child process of its own")
    config._fastf_units = {}          # unit key -> result dict
    config._fastf_unit_log = []


def _is_gpu(item):
    return item.get_closest_marker("gpu") is not None


def _unit_of(item):
    if item.get_closest_marker("own_process") is not None:
        return item.nodeid
    return item.nodeid.split("::")[0]


def pytest_collection_modifyitems(session, config, items):
    def key(it):
        mod = os.path.splitext(os.path.basename(it.nodeid.split("::")[0]))[0]
        return ORDER.index(mod) if mod in ORDER else len(ORDER)
    gpu = [it for it in items if _is_gpu(it)]
    if gpu and not CHILD:
        idx = {id(it): i for i, it in enumerate(items)}
        items.sort(key=lambda it: (key(it) if _is_gpu(it) else -1, idx[id(it)]))


# ---------------------------------------------------------------- child side: report every phase
def pytest_runtest_logreport(report):
    path = getattr(pytest, "_fastf_report_path", None)
    if not path:
        return
    rec = dict(nodeid=report.nodeid, when=report.when, outcome=report.outcome, duration=float(getattr(report, "duration", 0.0)),
               longrepr=str(report.longrepr) if report.longrepr else "",
               skipped=list(report.longrepr) if report.outcome == "skipped" and isinstance(report.longrepr, tuple) else None)
    with open(path, "a") as f:
        f.write(json.dumps(rec, default=str) + "\n")


def pytest_sessionstart(session):
    pytest._fastf_report_path = session.config.getoption("--fastf-report")


# ---------------------------------------------------------------- parent side: run a unit in a child, replay its reports
def _say(config, line):
    """one line on the terminal right now (a unit takes minutes: the log must show that the run is alive)"""
    tr = config.pluginmanager.get_plugin("terminalreporter")
    cap = config.pluginmanager.get_plugin("capturemanager")
    try:
        if cap is not None:
            with cap.global_and_fixture_disabled():
                (tr.write_line(line) if tr is not None else sys.stderr.write(line + "\n"))
                sys.stdout.flush()
        else:
            sys.stderr.write(line + "\n")
    except Exception:
        sys.stderr.write(line + "\n")


def _run_unit(config, unit, nodeids):
    tmp = tempfile.mkdtemp(prefix="fastf_unit_")
    rep, log = os.path.join(tmp, "report.jsonl"), os.path.join(tmp, "output.txt")
    env = dict(os.environ, FASTF_TEST_CHILD="1", PYTHONPATH=os.pathsep.join([ROOT] + ([os.environ["PYTHONPATH"]] if os.environ.get("PYTHONPATH") else [])))
    cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "--no-header", "-m", "gpu", "--fastf-report", rep] + list(nodeids)
    t0 = time.time()
    rc, note = None, ""
    with open(log, "wb") as lf:
        p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=lf, stderr=subprocess.STDOUT, start_new_session=True)
        try:
            rc = p.wait(timeout=UNIT_TIMEOUT_S)
        except subprocess.TimeoutExpired:
            note = "timed out after %d s" % UNIT_TIMEOUT_S
            try:
                os.killpg(p.pid, 9)
            except OSError:
                pass
            rc = p.wait()
    phases = {}
    if os.path.exists(rep):
        with open(rep) as f:
            for ln in f:
                try:
                    r = json.loads(ln)
                except ValueError:
                    continue
                phases.setdefault(r["nodeid"], []).append(r)
    with open(log, "rb") as f:
        out = f.read().decode("utf-8", "replace")
    res = dict(rc=rc, note=note, phases=phases, tail=out[-6000:], seconds=time.time() - t0, n=len(nodeids))
    done = [n for n in nodeids if any(ph["when"] == "teardown" for ph in phases.get(n, []))]
    bad = [n for n in done if any(ph["outcome"] == "failed" for ph in phases[n])]
    line = "[gpu unit] %-60s rc=%s %d/%d tests reported, %d failed, %.1f s%s" % (unit[-60:], rc, len(done), len(nodeids), len(bad), res["seconds"], (" — " + note) if note else "")
    if rc not in (0, 1) or len(done) < len(nodeids):
        fault = [ln for ln in out.splitlines() if "Memory access fault" in ln or "Fatal Python error" in ln or "HSA_STATUS" in ln]
        line += "  CHILD DIED" + (": " + fault[0][:200] if fault else "")
    config._fastf_unit_log.append(line)
    _say(config, line)
    # raw record of a unit that did not end cleanly (kept: profiles/ gets a copy by hand; gpurun_out/ travels back by itself)
    if rc not in (0, 1) or note:
        try:
            d = os.path.join(ROOT, "gpurun_out", "unit_failures"); os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "%d_%s.txt" % (int(t0), os.path.basename(unit).replace("/", "_")[:80])), "w") as f:
                f.write(line + "\n" + out)
        except OSError:
            pass
    return res


def _replay(item, res):
    from _pytest.reports import TestReport
    phases = res["phases"].get(item.nodeid, [])
    ihook = item.ihook
    ihook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
    complete = any(ph["when"] == "teardown" for ph in phases)
    for ph in phases:
        longrepr = ph["longrepr"] or None
        if ph["outcome"] == "skipped":
            sk = ph.get("skipped")
            longrepr = tuple(sk) if sk and len(sk) == 3 else (str(item.fspath), 0, ph["longrepr"] or "skipped")
        ihook.pytest_runtest_logreport(report=TestReport(item.nodeid, item.location, dict(item.keywords), ph["outcome"], longrepr, ph["when"], duration=ph["duration"]))
    if not complete:
        if res.get("died_at") and item.nodeid not in res["died_at"] and not phases:
            res = dict(res, tail=res.get("tail_more", res["tail"]))
        why = "the child process of this test's unit ended (rc=%s%s) before this test reported%s\n---- end of the child's output ----\n%s" % (
            res["rc"], (", " + res["note"]) if res["note"] else "", " its teardown" if phases else "", res["tail"])
        when = "call" if not any(ph["when"] == "call" for ph in phases) else "teardown"
        if not phases:
            ihook.pytest_runtest_logreport(report=TestReport(item.nodeid, item.location, dict(item.keywords), "passed", None, "setup"))
        ihook.pytest_runtest_logreport(report=TestReport(item.nodeid, item.location, dict(item.keywords), "failed", why, when))
        if when == "call":
            ihook.pytest_runtest_logreport(report=TestReport(item.nodeid, item.location, dict(item.keywords), "passed", None, "teardown"))
    ihook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)


@pytest.hookimpl(tryfirst=True)
def pytest_runtest_protocol(item, nextitem):
    if CHILD or INPROCESS or not _is_gpu(item):
        return None
    config = item.config
    if not config._fastf_units:
        # every unit runs before the first report is replayed: with -x the session ends at the first failed test, and the
        # units behind it would never have run — their one-line results are in the log (and in the summary) either way
        units = {}
        for it in item.session.items:
            if _is_gpu(it):
                units.setdefault(_unit_of(it), []).append(it.nodeid)
        for unit, nodeids in units.items():
            res = _run_unit(config, unit, nodeids)
            # a child that died took the test it was running with it — the tests behind that one get a fresh child (twice at most)
            for _ in range(2):
                seen = [n for n in nodeids if res["phases"].get(n)]
                if res["rc"] in (0, 1) or not seen or res["note"]:
                    break
                rest = nodeids[nodeids.index(seen[-1]) + 1:]
                if not rest:
                    break
                more = _run_unit(config, unit + " (after the child died)", rest)
                res["phases"].update(more["phases"])
                res["rc"], res["tail_more"] = more["rc"], more["tail"]
                res["died_at"] = res.get("died_at", []) + [seen[-1]]
            config._fastf_units[unit] = res
    _replay(item, config._fastf_units[_unit_of(item)])
    return True


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if getattr(config, "_fastf_unit_log", None):
        terminalreporter.section("GPU units (one child process each)")
        for ln in config._fastf_unit_log:
            terminalreporter.write_line(ln)
