"""CPU suite: the one-command N-rank launchers (bench.py --gpus N without ranks from outside; GAIB_RANKS=N
bin/gpu_train_*).  No GPU here, so the ranks are stand-in scripts (bench.py's launcher takes the rank program as an
argument for exactly this) or fail at their first GPU call -- what is checked is the supervision: rank 0's JSON line is
relayed last, the first failing rank ends the job with a non-zero status, the deadline ends a job that hangs, and no
rank process is left behind (one entry point drives all ranks, /root/reference/src/triangle/multigpu_induced.cu:31-84)."""
import json
import os
import subprocess
import sys
import textwrap
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _args(n, deadline=60.0):
    import argparse

    return argparse.Namespace(gpus=n, deadline_s=deadline)


def _alive(pid: int) -> bool:
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    # a zombie of a session we do not parent would still count; the launcher reaps its own children
    try:
        return open(f"/proc/{pid}/stat").read().split()[2] != "Z"
    except FileNotFoundError:
        return False


def _run_launcher(tmp_path, body, n, deadline=60.0):
    """run bench.launch_ranks in a child interpreter (it installs signal handlers) on a stand-in rank program"""
    rank_prog = tmp_path / "rank.py"
    rank_prog.write_text(textwrap.dedent(body))
    driver = tmp_path / "drive.py"
    driver.write_text(textwrap.dedent(f"""
        import argparse, sys
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        rc = bench.launch_ranks(argparse.Namespace(gpus={n}, deadline_s={deadline}), [{str(tmp_path)!r}], entry={str(rank_prog)!r})
        sys.exit(rc)
    """))
    t0 = time.time()
    r = subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=120)
    return r, time.time() - t0


def test_launcher_relays_rank0_json_last_and_sets_rank_environment(tmp_path):
    body = """
        import json, os, sys
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1"
        assert int(os.environ["MASTER_PORT"]) > 0 and os.environ["GAIB_LAUNCH_NONCE"]
        open(os.path.join(sys.argv[1], f"pid{r}"), "w").write(str(os.getpid()))
        print(f"chatter from rank {r}")
        if r == 0:
            print(json.dumps({"value": 1.0, "n_gpus": w}))
    """
    r, _ = _run_launcher(tmp_path, body, 3)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert json.loads(lines[-1]) == {"value": 1.0, "n_gpus": 3}
    assert len(lines) == 1  # chatter of every rank went to stderr
    assert "chatter from rank 0" in r.stderr and "chatter from rank 2" in r.stderr
    assert sorted(p.name for p in tmp_path.glob("pid*")) == ["pid0", "pid1", "pid2"]


def test_launcher_first_failing_rank_ends_the_job(tmp_path):
    body = """
        import os, sys, time
        r = int(os.environ["RANK"])
        open(os.path.join(sys.argv[1], f"pid{r}"), "w").write(str(os.getpid()))
        if r == 1:
            time.sleep(0.5)
            sys.exit(7)
        time.sleep(300)  # a rank waiting in a collective for the dead peer
    """
    r, took = _run_launcher(tmp_path, body, 3)
    assert r.returncode == 7 and took < 30, (r.returncode, took, r.stderr)
    assert "rank 1 exited with 7" in r.stderr
    assert r.stdout.strip() == ""
    for f in tmp_path.glob("pid*"):
        assert not _alive(int(f.read_text())), f"rank process {f.name} survived the launcher"


def test_launcher_deadline_ends_a_hanging_job(tmp_path):
    body = """
        import os, sys, time, signal
        r = int(os.environ["RANK"])
        open(os.path.join(sys.argv[1], f"pid{r}"), "w").write(str(os.getpid()))
        if r == 0:
            signal.signal(signal.SIGTERM, signal.SIG_IGN)  # a rank that ignores the first signal is still killed
        time.sleep(300)
    """
    r, took = _run_launcher(tmp_path, body, 2, deadline=2.0)
    assert r.returncode == 124 and took < 40, (r.returncode, took, r.stderr)
    assert "deadline" in r.stderr
    for f in tmp_path.glob("pid*"):
        assert not _alive(int(f.read_text()))


def test_rank_guard_prints_the_held_record_when_the_deadline_cuts_a_sub_case(tmp_path):
    """the N > 1 record cannot be lost: rank 0 holds the headline record, a later sub-case hangs (here: sleeps in C), the
    rank's own deadline fires -> rank 0 prints the held record marked partial and exits 0, the other rank exits 124, the
    launcher relays the line and reports success"""
    body = f"""
        import os, sys, time
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        r = int(os.environ["RANK"])
        open(os.path.join(sys.argv[1], f"pid{{r}}"), "w").write(str(os.getpid()))
        g = bench.install_rank_guard(r, 2.0)
        if r == 0:
            g.hold({{"value": 42.0, "n_gpus": 2, "config": {{"random_order": None}}}})
        time.sleep(300)  # the sub-case that never returns
    """
    r, took = _run_launcher(tmp_path, body, 2, deadline=60.0)
    assert r.returncode == 0 and took < 30, (r.returncode, took, r.stderr)
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["value"] == 42.0 and "deadline" in rec["partial"]["reason"]
    for f in tmp_path.glob("pid*"):
        assert not _alive(int(f.read_text()))


def test_rank_guard_prints_the_held_record_on_sigterm(tmp_path):
    """... and when the launcher (or the driver) ends the run from outside: SIGTERM reaches a rank that sits in a C call,
    a watcher thread -- not a Python signal handler -- prints what is held"""
    body = f"""
        import ctypes, os, sys, time
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        r = int(os.environ["RANK"])
        open(os.path.join(sys.argv[1], f"pid{{r}}"), "w").write(str(os.getpid()))
        g = bench.install_rank_guard(r, 500.0)
        if r == 0:
            g.hold({{"value": 7.0, "n_gpus": 2, "config": {{}}}})
        ctypes.CDLL(None).sleep(300)  # blocked inside C: no bytecode runs, a Python-level handler would never fire
    """
    r, took = _run_launcher(tmp_path, body, 2, deadline=2.0)
    assert r.returncode == 0 and took < 30, (r.returncode, took, r.stderr)
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["value"] == 7.0 and "signal" in rec["partial"]["reason"]
    for f in tmp_path.glob("pid*"):
        assert not _alive(int(f.read_text()))


def test_rank0_failing_in_a_sub_case_prints_the_held_record_and_the_job_ends_soon(tmp_path):
    """an exception on rank 0 after the headline case (bench.py catches it and calls bail from the main thread): the held record
    goes out marked partial, rank 0 exits 0, and the launcher does not wait the whole deadline for the ranks that sit in a
    collective rank 0 will never join -- 20 s after rank 0 left they are stopped, the line is relayed, status 0"""
    body = f"""
        import ctypes, os, sys, time
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        r = int(os.environ["RANK"])
        open(os.path.join(sys.argv[1], f"pid{{r}}"), "w").write(str(os.getpid()))
        g = bench.install_rank_guard(r, 500.0)
        if r == 0:
            g.hold({{"value": 9.0, "n_gpus": 2, "config": {{}}}})
            try:
                raise MemoryError("config 5 did not fit")
            except Exception as e:
                g.bail(f"{{type(e).__name__}}: {{e}}")
        ctypes.CDLL(None).sleep(300)  # rank 1: the collective that never completes
    """
    r, took = _run_launcher(tmp_path, body, 2, deadline=400.0)
    assert r.returncode == 0 and took < 60, (r.returncode, took, r.stderr)
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["value"] == 9.0 and "MemoryError" in rec["partial"]["reason"]
    for f in tmp_path.glob("pid*"):
        assert not _alive(int(f.read_text()))


def test_rank_guard_waits_a_moment_for_a_record_that_is_about_to_be_held(tmp_path):
    """a peer fails in the instant between the headline case's last collective and rank 0's hold(): the launcher's SIGTERM
    reaches rank 0 first.  bail() gives the main thread up to 5 s to hand the record over (found by the GPU test with an
    injected failure on rank 1: one run in three ended without a line)"""
    body = f"""
        import os, sys, time
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        r = int(os.environ["RANK"])
        open(os.path.join(sys.argv[1], f"pid{{r}}"), "w").write(str(os.getpid()))
        g = bench.install_rank_guard(r, 500.0)
        if r == 1:
            sys.exit(124)      # the failing peer: the launcher now stops rank 0
        time.sleep(1.5)        # rank 0 is still assembling its record when the signal arrives ...
        g.hold({{"value": 5.0, "n_gpus": 2, "config": {{}}}})
        time.sleep(300)
    """
    r, took = _run_launcher(tmp_path, body, 2, deadline=400.0)
    assert r.returncode == 0 and took < 60, (r.returncode, took, r.stderr)
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["value"] == 5.0 and "signal" in rec["partial"]["reason"]


def test_launcher_relays_the_held_record_when_it_is_signalled_itself(tmp_path):
    """the driver's timeout ends `python bench.py --gpus N` with SIGTERM: the launcher stops its ranks, rank 0 prints the record
    it holds, and the launcher puts that line on its own stdout before it leaves (it used to exit with the line still in its pipe)"""
    import signal

    rank_prog = tmp_path / "rank.py"
    rank_prog.write_text(textwrap.dedent(f"""
        import ctypes, os, sys
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        r = int(os.environ["RANK"])
        open(os.path.join(sys.argv[1], f"pid{{r}}"), "w").write(str(os.getpid()))
        g = bench.install_rank_guard(r, 500.0)
        if r == 0:
            g.hold({{"value": 11.0, "n_gpus": 2, "config": {{}}}})
        ctypes.CDLL(None).sleep(300)
    """))
    driver = tmp_path / "drive.py"
    driver.write_text(textwrap.dedent(f"""
        import argparse, sys
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        sys.exit(bench.launch_ranks(argparse.Namespace(gpus=2, deadline_s=400.0), [{str(tmp_path)!r}], entry={str(rank_prog)!r}))
    """))
    p = subprocess.Popen([sys.executable, str(driver)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while time.time() - t0 < 30 and len(list(tmp_path.glob("pid*"))) < 2:
        time.sleep(0.1)
    time.sleep(1.0)  # (rank 0 has held its record by now)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err)
    rec = json.loads(out.strip().splitlines()[-1])
    assert rec["value"] == 11.0 and "signal" in rec["partial"]["reason"]
    for f in tmp_path.glob("pid*"):
        assert not _alive(int(f.read_text()))


def test_record_survives_sigterm_when_native_threads_exist_before_the_guard(tmp_path):
    """threads created BEFORE the signal mask (torch's OpenMP pool, HSA's event thread) keep SIGTERM unblocked at
    SIG_DFL, the kernel hands a process-directed SIGTERM to one of them and the process dies without a record.  bench.main()
    therefore blocks the signals as its first act in a rank process, before `import torch`; this program keeps that order
    (mask -> torch + a matmul that spins up the pool -> guard -> hold -> C sleep) and must print the held record"""
    import signal

    prog = tmp_path / "rank.py"
    prog.write_text(textwrap.dedent(f"""
        import ctypes, os, signal, sys
        signal.pthread_sigmask(signal.SIG_BLOCK, {{signal.SIGTERM, signal.SIGINT}})   # what bench.main() does first
        sys.path.insert(0, {str(ROOT)!r})
        import torch
        torch.set_num_threads(8)
        a = torch.randn(1024, 1024)
        (a @ a).sum().item()                      # native worker threads exist from here on
        import threading
        n_native = len(os.listdir("/proc/self/task"))
        import bench
        g = bench.install_rank_guard(0, 500.0)
        g.want_parity = True
        g.hold({{"value": 13.0, "n_gpus": 1, "config": {{}}, "threads_before_guard": n_native}})
        open(os.path.join(sys.argv[1], "ready"), "w").write(str(os.getpid()))
        ctypes.CDLL(None).sleep(300)
    """))
    p = subprocess.Popen([sys.executable, str(prog), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while time.time() - t0 < 120 and not (tmp_path / "ready").exists():
        time.sleep(0.1)
    assert (tmp_path / "ready").exists(), p.stderr.read() if p.poll() is not None else "rank program did not get ready"
    time.sleep(0.3)
    p.send_signal(signal.SIGTERM)  # process-directed, like the launcher's / the driver's
    out, err = p.communicate(timeout=60)
    assert p.returncode == 0, (p.returncode, err)
    rec = json.loads(out.strip().splitlines()[-1])
    assert rec["value"] == 13.0 and "signal" in rec["partial"]["reason"]
    assert rec["threads_before_guard"] > 1  # the situation the finding describes was really there
    assert rec["parity"]["ok"] is None and "did not complete" in rec["parity"]["reason"]


def test_bench_main_blocks_the_signals_before_it_imports_torch():
    """the order in bench.main() itself (it cannot run here without a GPU): pthread_sigmask + the guard come before `import torch`"""
    import inspect

    import bench

    src = inspect.getsource(bench.main)
    i_mask, i_guard, i_torch = src.index("signal.pthread_sigmask(signal.SIG_BLOCK"), src.index("install_rank_guard("), src.index("import torch")
    assert i_mask < i_guard < i_torch
    assert src.index("launch_ranks(args") < i_mask  # the launcher parent keeps its signal HANDLERS: it never blocks


def test_a_crashing_leg_after_the_headline_prints_the_record_and_exits_5(tmp_path):
    """a leg that crashes after the measurement must not look like a passing run by exit status -- the held record
    goes out (relayed by the launcher), the status is 5"""
    body = f"""
        import ctypes, os, sys
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        r = int(os.environ["RANK"])
        open(os.path.join(sys.argv[1], f"pid{{r}}"), "w").write(str(os.getpid()))
        g = bench.install_rank_guard(r, 500.0)
        g.want_parity = True
        if r == 0:
            g.hold({{"value": 9.5, "n_gpus": 2, "config": {{}}}})
            g.bail("RuntimeError: the parity leg crashed", status=5)
        ctypes.CDLL(None).sleep(300)
    """
    r, took = _run_launcher(tmp_path, body, 2, deadline=400.0)
    assert r.returncode == 5 and took < 60, (r.returncode, took, r.stderr)
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["value"] == 9.5 and rec["parity"]["ok"] is None and "crashed" in rec["partial"]["reason"]
    for f in tmp_path.glob("pid*"):
        assert not _alive(int(f.read_text()))


def test_rank_guard_emits_exactly_one_line(tmp_path):
    """final() after hold(): one line, not marked partial; a bail() that races with it prints nothing more"""
    prog = tmp_path / "one.py"
    prog.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        g = bench.install_rank_guard(0, 500.0)
        g.hold({{"value": 1.0}})
        g.final({{"value": 2.0}})
        g.bail("late")  # returns: the record is out
        g.final({{"value": 3.0}})
    """))
    r = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert [json.loads(l) for l in r.stdout.strip().splitlines()] == [{"value": 2.0}]


def test_budget_skips_sub_cases_collectively():
    """dist.Budget: a sub-case starts only if EVERY rank still has its estimated time (all-reduce(MIN)); a skipped slot says so"""
    from graphaibench_amd.dist import Budget

    now = [100.0]
    votes = []

    def reduce_min(flag):  # rank B of a two-rank run is always 30 s behind
        other = 1 if (400.0 - (now[0] + 30.0 - 100.0)) >= votes[-1] else 0
        return min(flag, other)

    b = Budget(400.0, 100.0, reduce_min, clock=lambda: now[0])
    votes.append(50.0)
    assert b.agree(50.0)  # 400 s left here, 370 on the other rank
    now[0] = 100.0 + 340.0
    votes.append(50.0)
    assert b.left() == 60.0 and not b.agree(50.0)  # this rank could (60 s left), the other cannot (30 s): nobody runs it
    sk = b.skipped(50.0)
    assert sk == {"skipped": "budget", "elapsed_s": 340.0, "budget_s": 400.0, "needed_s_estimate": 50.0}
    votes.append(500.0)
    assert not b.agree(500.0)


def test_bench_plain_invocation_without_gpu_fails_fast_and_clean():
    """`python bench.py --gpus 2` with a clean environment on a box without a GPU: the parent starts the ranks, they
    refuse to run without a GPU (no CPU fallback), the launcher reports it and exits non-zero -- no hang, no JSON"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    import torch

    if torch.cuda.is_available():
        pytest.skip("the GPU suite runs the real thing (tests/test_gpu_dist.py)")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--scale", "0.01"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and "exited with" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_rejects_mismatched_world_and_gat_multi_gpu():
    env = dict(os.environ, RANK="0", WORLD_SIZE="3", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr
    # (round 6: gat-reddit runs across ranks; the epoch workloads -- the trainer CLI -- stay one-GPU records)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "epoch-gcn-cora"], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 2 and "one-GPU workload" in r.stderr


def test_trainer_launcher_stops_all_ranks_when_one_fails():
    """GAIB_RANKS=3 bin/gpu_train_gcn on a box without a GPU: every rank dies at its first GPU call; the launcher says
    which rank, exits with its status and leaves nothing behind"""
    import torch

    exe = ROOT / "bin" / "gpu_train_gcn"
    if not exe.exists():
        pytest.skip("trainer not built")
    if torch.cuda.is_available():
        pytest.skip("the GPU suite runs the real thing (tests/test_gpu_driver.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GAIB_RANK")}
    env.update(GAIB_RANKS="3", DATASET_PATH=str(ROOT / "tests" / "golden") + "/")
    r = subprocess.run([str(exe), "cora", "1", "1", "softmax"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0
    assert "[launcher] rank" in r.stderr and "stopping the other ranks" in r.stderr


def test_epoch_record_helpers():
    """bench.py's epoch workloads: the 128-B lines a gathered row touches (the physical floor next to SURVEY 8(d)'s algorithmic
    bytes) and the parser of the trainer's "[gaib prof]" table"""
    import bench
    from graphaibench_amd import capi

    assert bench._lines_per_row(47) == 2.0 and bench._lines_per_row(128) == 4.0 and bench._lines_per_row(32) == 1.0
    assert bench._lines_per_row(100) == 4.0 and bench._lines_per_row(16) == 1.0 and bench._lines_per_row(256) == 8.0
    tab = capi.parse_prof_table("spmm_gemm_fused@128 4 30.5 2.3e11 1.2e11 28.75\nsgemm 6 7.25 1e10 9e11 5.7\n\nnot a line\n")
    assert tab["spmm_gemm_fused@128"] == dict(count=4, ms=30.5, bytes=2.3e11, flops=1.2e11, roof_ms=28.75) and tab["sgemm"]["count"] == 6
    assert set(bench.EPOCH_WORKLOADS) == {"epoch-sage-products", "epoch-gcn-products", "epoch-gat-reddit", "epoch-gcn-cora"}


def test_epoch_record_is_assembled_from_the_trainer_output(monkeypatch, tmp_path):
    """bench_epoch with the trainer and the dataset writer replaced by canned outputs (no GPU here): the record's arithmetic --
    value from the trainer's own epoch times and edge count, the fraction = the launches' roof time over the epoch, the
    line floor only next to gather kernels whose rows are no whole number of lines, dense products listed per shape"""
    import argparse

    import bench

    class FakeTorch:
        class cuda:
            @staticmethod
            def empty_cache():
                pass

    class FakeSynth:
        @staticmethod
        def write_dataset(name, root, scale=1.0, device="cuda"):
            return dict(nv=1000, ne=5_000_000, F=100, C=47, train_begin=0, train_end=80, max_degree=99, dir=str(tmp_path / name))  # (> 4 M edges: not the launch-bound two-run scheme)

    table = ("spmm_gemm_fused@128 4 20.0 1.2e11 4e10 16.0\nspmm_light@47 2 6.0 3.0e10 1e9 4.0\nsgemm@1000x128x100 6 3.0 6e9 1.2e11 1.5\n"
             "d_relu 2 0.1 1e8 0 0.0125\n")

    def fake_trainer(arch, data_root, dataset, epochs, hidden, layers, heads, prof_from, timeout_s, times_from=None):
        ep = [dict(loss=3.8 - 0.1 * i, acc=0.02 * i, seconds=0.020 if i else 0.5) for i in range(epochs)]
        return "", ep, bench.__dict__["capi_parse"](table) if prof_from is not None else {}, 250000

    from graphaibench_amd import capi
    monkeypatch.setitem(bench.__dict__, "capi_parse", capi.parse_prof_table)
    monkeypatch.setattr(bench, "_run_trainer", fake_trainer)
    held = {}

    class Guard:
        held = None

        def hold(self, r):
            held["r"] = r

        def final(self, r):
            held["final"] = r

    args = argparse.Namespace(workload="epoch-gcn-products", steps=2, warmup=1, scale=1.0, no_cpu_baseline=True)
    assert bench.bench_epoch(args, FakeTorch, FakeSynth, Guard()) == 0
    r = held["final"]
    assert r["ms_per_step"] == pytest.approx(20.0) and r["value"] == pytest.approx(250000 * 2 / 0.040) and r["steps"] == 2
    roof = r["roofline"]
    assert roof["roof_ms_per_epoch"] == pytest.approx((16.0 + 4.0 + 1.5 + 0.0125) / 2) and roof["frac"] == pytest.approx(roof["roof_ms_per_epoch"] / 20.0)
    assert roof["untimed_ms_per_epoch"] == pytest.approx(20.0 - 29.1 / 2)
    pk = roof["per_key"]
    assert "line_floor" in pk["spmm_light@47"] and pk["spmm_light@47"]["line_floor"]["lines_per_row"] == 2.0
    assert "line_floor" not in pk["spmm_gemm_fused@128"] and "line_floor" not in pk["sgemm@1000x128x100"] and "line_floor" not in pk["d_relu"]
    assert pk["sgemm@1000x128x100"]["frac"] == pytest.approx(0.5) and r["config"]["train_loss_timed_epochs"] == [pytest.approx(3.7), pytest.approx(3.6)]


def test_child_programs_run_unmasked_and_do_not_outlive_a_bailing_guard(tmp_path):
    """bench.py blocks SIGTERM / SIGINT first thing and the mask survives fork + exec -- the trainer child of
    an epoch leg must start with the mask RESTORED (it has to die on the pool's SIGTERM) and must not be left on the GPU when
    RecordGuard.bail() leaves through os._exit: run_child starts it in its own session through the unmasking wrapper, bail()
    kills and reaps the group first"""
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(f"""
        import os, signal, sys, time
        blocked = signal.pthread_sigmask(signal.SIG_BLOCK, [])
        open({str(tmp_path / "child_info")!r}, "w").write(f"{{os.getpid()}} {{int(signal.SIGTERM in blocked)}} {{int(signal.SIGINT in blocked)}}")
        time.sleep(300)
    """))
    parent = tmp_path / "parent.py"
    parent.write_text(textwrap.dedent(f"""
        import signal, sys, threading
        signal.pthread_sigmask(signal.SIG_BLOCK, {{signal.SIGTERM, signal.SIGINT}})  # bench.main()'s first act
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        g = bench.install_rank_guard(0, 3.0)
        g.hold({{"value": 7.0, "config": {{}}}})
        try:
            rc, _, _ = bench.run_child([sys.executable, {str(child)!r}], timeout_s=120)  # "the epoch leg": never returns by itself
            raise RuntimeError(f"the trainer exited with {{rc}}")  # (what _run_trainer makes of a killed child)
        except Exception as e:
            g.bail(str(e))  # main()'s handler: must not outrun the watcher thread that is printing the record
            raise
    """))
    t0 = time.time()
    r = subprocess.run([sys.executable, str(parent)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and time.time() - t0 < 30, (r.returncode, r.stderr[-2000:])
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["value"] == 7.0 and "deadline" in rec["partial"]["reason"]
    pid, term_blocked, int_blocked = (int(v) for v in (tmp_path / "child_info").read_text().split())
    assert term_blocked == 0 and int_blocked == 0, "the child inherited the blocked signals"
    assert not _alive(pid), "the child program outlived the bailing guard"
    assert "killed 1 child program" in r.stderr


def test_run_child_returns_output_and_kills_on_timeout(tmp_path):
    import bench

    ok = tmp_path / "ok.py"
    ok.write_text("import sys; print('out'); print('err', file=sys.stderr); sys.exit(3)")
    rc, out, err = bench.run_child([sys.executable, str(ok)], timeout_s=30)
    assert (rc, out.strip(), err.strip()) == (3, "out", "err") and not bench._CHILDREN
    slow = tmp_path / "slow.py"
    slow.write_text(f"import os, time; open({str(tmp_path / 'slow_pid')!r}, 'w').write(str(os.getpid())); time.sleep(300)")
    with pytest.raises(subprocess.TimeoutExpired):
        bench.run_child([sys.executable, str(slow)], timeout_s=2)
    assert not _alive(int((tmp_path / "slow_pid").read_text()))


def test_other_configs_block_is_budgeted_and_failure_proof(monkeypatch, tmp_path):
    """the default N = 1 run's `other_configs` slot: every leg present when the budget allows, {"skipped": "budget"}
    when it does not, {"error": ...} when a leg raises -- the block never raises; the epoch legs' records are cut down to the
    slot's fields; the products dataset is written once for the three models on it"""
    import argparse

    import bench
    from graphaibench_amd import capi

    class FakeTorch:
        class cuda:
            @staticmethod
            def empty_cache():
                pass

    writes = []

    class FakeSynth:
        @staticmethod
        def write_dataset(name, root, scale=1.0, device="cuda"):
            writes.append(name)
            d = Path(root) / name
            d.mkdir(parents=True, exist_ok=True)
            return dict(nv=1000, ne=5_000_000, F=100, C=47, train_begin=0, train_end=80, max_degree=99, dir=str(d))

        @staticmethod
        def write_cora_dataset(root, golden):
            writes.append("cora")
            d = Path(root) / "cora"
            d.mkdir(parents=True, exist_ok=True)
            return dict(nv=2708, ne=13264, F=1433, C=7, train_begin=0, train_end=140, max_degree=168, dir=str(d))

    table = "spmm_gemm_fused@128 4 20.0 1.2e11 4e10 16.0\nspmm_light@47 2 6.0 3.0e10 1e9 4.0\nsgemm@1000x128x100 6 3.0 6e9 1.2e11 1.5\n"
    hiddens = []

    def fake_trainer(arch, data_root, dataset, epochs, hidden, layers, heads, prof_from, timeout_s, times_from=None):
        hiddens.append((arch, dataset, hidden))
        ep = [dict(loss=3.8 - 0.1 * i, acc=0.02 * i, seconds=0.020 if i else 0.5) for i in range(epochs)]
        return "", ep, capi.parse_prof_table(table) if prof_from is not None else {}, 250000

    monkeypatch.setattr(bench, "_run_trainer", fake_trainer)
    monkeypatch.setattr(bench, "sage_layer_step", lambda torch, ctx, L, sg, width, steps, warmup: {"ms_per_step": float(width), "value": 1.0})

    def fake_gat(a2, torch, ctx, L, synth, cap):
        if a2.scale == 0.5:
            raise RuntimeError("boom")
        cap.hold({"config": {"workload": "gat", "nv": 1, "ne_with_selfloops": 2, "heads": 8}, "value": 3.0, "unit": "edges/s", "ms_per_step": 8.6,
                  "roofline": None, "breakdown_ms_per_step": {}})

    monkeypatch.setattr(bench, "bench_gat_reddit", fake_gat)
    updates = []
    args = argparse.Namespace(scale=1.0, steps=20, warmup=3, other_configs_s=100.0, no_cpu_baseline=False, workload="gcn-products")
    out = bench.other_configs(args, FakeTorch, None, None, FakeSynth, None, time.time() + 1000, lambda o: updates.append(dict(o)))
    legs = ["sage_layer_128", "sage_layer_256", "gat_layer_reddit_8x8", "epoch_sage_products_hidden256", "epoch_sage_products_hidden128",
            "epoch_gcn_products", "epoch_gat_reddit", "epoch_gcn_cora"]
    assert all(k in out for k in legs) and len(updates) == len(legs)
    assert out["sage_layer_256"]["ms_per_step"] == 256.0 and out["gat_layer_reddit_8x8"]["ms_per_step"] == 8.6
    e = out["epoch_sage_products_hidden128"]
    assert e["hidden"] == 128 and e["ms_per_epoch"] == pytest.approx(20.0) and e["steps"] == 10 and 0 < e["roofline"]["frac"] < 1
    assert set(e["roofline"]["per_key"]) == {"spmm_gemm_fused@128", "spmm_light@47", "sgemm@1000x128x100"}
    assert "frac_of_line_floor" in e["roofline"]["per_key"]["spmm_light@47"]
    assert ("sage", "ogbn-products", 256) in hiddens and ("sage", "ogbn-products", 128) in hiddens and ("gcn", "ogbn-products", 128) in hiddens
    assert writes.count("ogbn-products") == 1 and writes.count("reddit") == 1  # one dataset for the three products models
    # no time left: every leg says so; a leg that raises costs only its own slot
    out = bench.other_configs(args, FakeTorch, None, None, FakeSynth, None, time.time() - 1, lambda o: None)
    assert all(out[k]["skipped"] == "budget" for k in legs)
    args.scale = 0.5
    out = bench.other_configs(args, FakeTorch, None, None, FakeSynth, None, time.time() + 1000, lambda o: None)
    assert "boom" in out["gat_layer_reddit_8x8"]["error"] and out["epoch_gcn_cora"]["ms_per_epoch"] > 0
