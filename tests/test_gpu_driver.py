"""GPU suite, driver level: bin/gpu_train_{gcn,sage,gat} -- binary dataset
reader -> model -> training loop with the reference's CLI and log lines -- on the cora topology
(tests/golden/cora, real reference data files) with seeded synthetic features, against an oracle
model composed from the CPU restatement (same Glorot seeds, same optimizer sharing quirks)."""
import os
import re
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import binding as orc

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"


def make_dataset(tmp_path, feat_len=96, seed=0):
    """$DATASET_PATH/cora/ with the reference's own topology/label files + a synthetic graph.feats.bin"""
    d = tmp_path / "data" / "cora"
    d.mkdir(parents=True)
    for f in ("graph.vertex.bin", "graph.edge.bin", "graph.vlabel.bin"):
        shutil.copyfile(GOLD / "cora" / f, d / f)
    meta = (GOLD / "cora" / "graph.meta.txt").read_text().split()
    meta[7] = str(feat_len)  # feat_len field (the shipped file says 1433 but ships no features)
    (d / "graph.meta.txt").write_text("\n".join(meta) + "\n")
    rng = np.random.default_rng(seed)
    labels = np.fromfile(d / "graph.vlabel.bin", np.uint8)
    x = rng.standard_normal((2708, feat_len)).astype(np.float32) * 0.5
    x[np.arange(2708), labels.astype(int) % feat_len] += 1.5  # learnable signal
    x.tofile(d / "graph.feats.bin")
    return str(tmp_path / "data") + "/", x, labels, [int(v) for v in meta[10:19]]


from oracle.model import OracleModel  # noqa: E402  (the reference's Model composed from the oracle's layers)


@pytest.mark.parametrize("arch,layers", [("gcn", 2), ("sage", 2), ("gat", 2), ("gcn", 3)])
def test_driver_loss_curve_tracks_oracle(tmp_path, arch, layers):
    root, x, labels, splits = make_dataset(tmp_path)
    tb, te = splits[0], splits[1]
    epochs, hid, lr = 6, 16, 0.01
    exe = ROOT / "bin" / f"gpu_train_{arch}"
    assert exe.exists(), "run graphaibench_amd.build"
    env = dict(os.environ, DATASET_PATH=root)
    cmd = [str(exe), "cora", str(epochs), "2", "softmax", str(hid), "0", "0", str(lr), str(layers), "0", "4", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = [(float(a), float(b)) for a, b in re.findall(r"train_loss ([0-9.]+) train_acc ([0-9.]+)", r.stdout)]
    assert len(got) == epochs, r.stdout
    assert "Test accuracy:" in r.stdout and "Average training time per epoch" in r.stdout
    assert "val_acc" in r.stdout  # val_interval = 4 -> evaluated at epoch 4

    rp = np.fromfile(GOLD / "cora" / "graph.vertex.bin", np.int64)
    ci = np.fromfile(GOLD / "cora" / "graph.edge.bin", np.uint32)
    masks = np.zeros(2708, np.uint8)
    masks[tb:te] = 1
    m = OracleModel(arch, rp, ci, x.shape[1], hid, 7, layers, lr)
    want = [m.epoch(x, labels, tb, te, masks) for _ in range(epochs)]
    for (gl, ga), (wl, wa) in zip(got, want):
        assert abs(gl - wl) < 2e-3, (got, want)
        assert abs(ga - wa) < 0.02, (got, want)
    assert want[-1][0] < want[0][0]  # it learns


@pytest.mark.parametrize("arch", ["gcn", "sage"])
def test_driver_sigmoid_loss_tracks_oracle(tmp_path, arch):
    """argv[4] = sigmoid: multi-hot labels (reader), sigmoid loss layer, micro-F1 as the accuracy"""
    root, x, labels, splits = make_dataset(tmp_path)
    tb, te = splits[0], splits[1]
    epochs, hid, lr = 6, 16, 0.01
    exe = ROOT / "bin" / f"gpu_train_{arch}"
    cmd = [str(exe), "cora", str(epochs), "2", "sigmoid", str(hid), "0", "0", str(lr), "2", "0", "4", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, DATASET_PATH=root), timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "multi-class (multi-hot) labels" in r.stdout
    got = [(float(a), float(b)) for a, b in re.findall(r"train_loss ([0-9.]+) train_acc ([0-9.]+)", r.stdout)]
    assert len(got) == epochs and "val_acc" in r.stdout and "Test accuracy:" in r.stdout
    rp = np.fromfile(GOLD / "cora" / "graph.vertex.bin", np.int64)
    ci = np.fromfile(GOLD / "cora" / "graph.edge.bin", np.uint32)
    masks = np.zeros(2708, np.uint8)
    masks[tb:te] = 1
    hot = np.zeros((2708, 7), np.uint8)
    hot[np.arange(2708), labels] = 1
    m = OracleModel(arch, rp, ci, x.shape[1], hid, 7, 2, lr)
    want = [m.epoch(x, hot, tb, te, masks, sigmoid=True) for _ in range(epochs)]
    for (gl, ga), (wl, wa) in zip(got, want):
        assert abs(gl - wl) < 2e-3, (got, want)
        assert abs(ga - wa) < 0.02, (got, want)
    assert want[-1][0] < want[0][0]


def test_driver_error_paths(tmp_path):
    exe = ROOT / "bin" / "gpu_train_gcn"
    r = subprocess.run([str(exe), "cora", "1", "1", "softmax"], capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k != "DATASET_PATH"})
    assert r.returncode != 0 and "DATASET_PATH" in r.stderr
    root, *_ = make_dataset(tmp_path)
    r = subprocess.run([str(exe), "nosuch", "1", "1", "softmax"], capture_output=True, text=True,
                       env=dict(os.environ, DATASET_PATH=root))
    assert r.returncode != 0 and "Failed to open file" in r.stderr


@pytest.mark.parametrize("arch", ["gcn", "sage", "gat"])
def test_driver_subgraph_sampling_and_inductive(tmp_path, arch):
    """GraphSAINT-style sampling (subg_size > 0) trains on sampled subgraphs of the
    training-masked graph (layers resized per epoch) and evaluates on the full graph; inductive=1
    without sampling trains on the masked graph."""
    root, x, labels, splits = make_dataset(tmp_path)
    meta = (tmp_path / "data" / "cora" / "graph.meta.txt").read_text().split()
    meta[10:13] = ["0", "1500", "1500"]  # a training range large enough to sample from
    (tmp_path / "data" / "cora" / "graph.meta.txt").write_text("\n".join(meta) + "\n")
    exe = ROOT / "bin" / f"gpu_train_{arch}"
    env = dict(os.environ, DATASET_PATH=root)
    for subg, inductive in [("600", "0"), ("0", "1")]:
        cmd = [str(exe), "cora", "12", "3", "softmax", "16", "0", "0", "0.02", "2", subg, "50", inductive]
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        losses = [float(a) for a in re.findall(r"train_loss ([0-9.]+)", r.stdout)]
        assert len(losses) == 12 and all(np.isfinite(losses))
        assert np.mean(losses[-3:]) < np.mean(losses[:3]), losses
        acc = float(re.search(r"Test accuracy: ([0-9.]+)", r.stdout).group(1))
        assert acc > 0.3, r.stdout[-1500:]  # 7 classes; the synthetic features carry the label


@pytest.mark.parametrize("arch,world", [("gcn", 2), ("sage", 3), ("gat", 2), ("gcn", 4), ("gat", 3)])  # (the box allows 6 GPU processes, pytest included: margin of one)
def test_driver_multi_rank_matches_single_rank(tmp_path, arch, world):
    """bin/gpu_train_* as N processes (one per rank; here all on cuda:0 over the IPC transport): vertex-range
    partition, halo exchange before every aggregation, gradient all-reduce before every optimizer step -- all behind
    the C ABI.  The loss curve of rank 0 equals the single-process run's; every rank exits 0."""
    root, x, labels, splits = make_dataset(tmp_path)
    epochs = 6
    exe = ROOT / "bin" / f"gpu_train_{arch}"
    cmd = [str(exe), "cora", str(epochs), "2", "softmax", "16", "0", "0", "0.01", "2", "0", "4", "0"]
    base = dict(os.environ, DATASET_PATH=root)
    single = subprocess.run(cmd, capture_output=True, text=True, env=base, timeout=600)
    assert single.returncode == 0, single.stdout[-2000:] + single.stderr[-2000:]
    want = re.findall(r"train_loss ([0-9.]+) train_acc ([0-9.]+)", single.stdout)
    want_test = float(re.search(r"Test accuracy: ([0-9.]+)", single.stdout).group(1))
    procs = []
    for r in range(world):
        env = dict(base, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", GAIB_DEVICE="0", GAIB_COMM="ipc",
                   GAIB_COMM_ID_FILE=str(tmp_path / "comm_id"), GAIB_COMM_TIMEOUT_S="60")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [(o[0][-1500:], o[1][-1500:]) for o in outs]
    got = re.findall(r"train_loss ([0-9.]+) train_acc ([0-9.]+)", outs[0][0])
    assert len(got) == epochs and "val_acc" in outs[0][0], outs[0][0]
    for (gl, ga), (wl, wa) in zip(got, want):
        assert abs(float(gl) - float(wl)) <= 2e-3 and abs(float(ga) - float(wa)) <= 0.01, (got, want)
    assert abs(float(re.search(r"Test accuracy: ([0-9.]+)", outs[0][0]).group(1)) - want_test) <= 0.01
    assert all(f"rank {r} of {world}: rows [" in outs[r][0] for r in range(world))
    assert "train_loss" not in outs[1][0]  # rank 0 prints the log lines


@pytest.mark.parametrize("arch,world", [("gcn", 3), ("sage", 2)])
def test_driver_one_command_launches_all_ranks(tmp_path, arch, world):
    """GAIB_RANKS=N bin/gpu_train_*: ONE command, no RANK / WORLD_SIZE from outside -- the process is the launcher
    (it touches no GPU API), starts N copies of itself, and rank 0's log equals the single-process run's.  With more
    ranks than devices (this box: one GPU) the ranks agree on the peer-to-peer transport by themselves."""
    root, x, labels, splits = make_dataset(tmp_path)
    exe = ROOT / "bin" / f"gpu_train_{arch}"
    cmd = [str(exe), "cora", "5", "2", "softmax", "16", "0", "0", "0.01", "2", "0", "4", "0"]
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GAIB_RANK", "GAIB_WORLD",
                                                                "GAIB_COMM", "GAIB_DEVICE", "GAIB_COMM_ID_FILE")}
    base = dict(clean, DATASET_PATH=root)
    single = subprocess.run(cmd, capture_output=True, text=True, env=base, timeout=600)
    assert single.returncode == 0, single.stdout[-2000:] + single.stderr[-2000:]
    want = re.findall(r"train_loss ([0-9.]+) train_acc ([0-9.]+)", single.stdout)
    multi = subprocess.run(cmd, capture_output=True, text=True, env=dict(base, GAIB_RANKS=str(world), GAIB_COMM_TIMEOUT_S="60"),
                           timeout=600)
    assert multi.returncode == 0, multi.stdout[-2000:] + multi.stderr[-2000:]
    got = re.findall(r"train_loss ([0-9.]+) train_acc ([0-9.]+)", multi.stdout)
    assert len(got) == 5 == len(want)
    for (gl, ga), (wl, wa) in zip(got, want):
        assert abs(float(gl) - float(wl)) <= 2e-3 and abs(float(ga) - float(wa)) <= 0.01, (got, want)
    assert all(f"rank {r} of {world}: rows [" in multi.stdout for r in range(world)), multi.stdout[:3000]
    if torch_device_count() < world:
        assert "GAIB_COMM=ipc" in multi.stderr
    # a rank that dies takes the job down with a non-zero status: an unreadable dataset on every rank
    bad = subprocess.run([str(exe), "nosuch", "1", "1", "softmax"], capture_output=True, text=True,
                         env=dict(base, GAIB_RANKS="2"), timeout=120)
    assert bad.returncode != 0 and "[launcher] rank" in bad.stderr


@pytest.mark.parametrize("arch,world", [("gcn", 3), ("sage", 2), ("gat", 2)])
def test_driver_ranks_over_the_rccl_branch(tmp_path, arch, world):
    """the same one-command launch with GAIB_COMM=rccl: every halo exchange, reverse exchange and gradient all-reduce of
    the trainer goes through comm.hip's RCCL branch, bound (GAIB_RCCL_LIB) to tests/fake_rccl's strict double because
    this box has one GPU -- rank 0's log equals the single-process run's"""
    fake = ROOT / "tests" / "fake_rccl" / "librccl_fake.so"
    assert fake.exists(), "python -m graphaibench_amd.build"
    root, x, labels, splits = make_dataset(tmp_path)
    exe = ROOT / "bin" / f"gpu_train_{arch}"
    cmd = [str(exe), "cora", "5", "2", "softmax", "16", "0", "0", "0.01", "2", "0", "4", "0"]
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GAIB_RANK", "GAIB_WORLD",
                                                                "GAIB_COMM", "GAIB_DEVICE", "GAIB_COMM_ID_FILE")}
    base = dict(clean, DATASET_PATH=root)
    single = subprocess.run(cmd, capture_output=True, text=True, env=base, timeout=600)
    assert single.returncode == 0, single.stdout[-2000:] + single.stderr[-2000:]
    want = re.findall(r"train_loss ([0-9.]+) train_acc ([0-9.]+)", single.stdout)
    multi = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                           env=dict(base, GAIB_RANKS=str(world), GAIB_COMM="rccl", GAIB_RCCL_LIB=str(fake), GAIB_COMM_TIMEOUT_S="60"))
    assert multi.returncode == 0, multi.stdout[-2000:] + multi.stderr[-2000:]
    assert "GAIB_COMM=ipc" not in multi.stderr and "fake_rccl:" not in multi.stderr, multi.stderr[-2000:]
    got = re.findall(r"train_loss ([0-9.]+) train_acc ([0-9.]+)", multi.stdout)
    assert len(got) == 5 == len(want)
    for (gl, ga), (wl, wa) in zip(got, want):
        assert abs(float(gl) - float(wl)) <= 2e-3 and abs(float(ga) - float(wa)) <= 0.01, (got, want)


@pytest.mark.parametrize("arch,how,world", [("gcn", "cm", 1), ("sage", "degree", 1), ("gat", "bfs", 1), ("gcn", "cm", 2)])
def test_driver_reordered_dataset_trains_the_same(tmp_path, arch, how, world):
    """GAIB_REORDER=cm|bfs|degree: the trainer relabels the dataset once on the host with the numbering gaib_graph_reorder
    computes (rows, features, labels, masks) -- the same model on the same graph, so the loss curve and the accuracies are
    the plain run's up to fp32 summation order, on one rank and on a partition"""
    root, x, labels, splits = make_dataset(tmp_path)
    exe = ROOT / "bin" / f"gpu_train_{arch}"
    cmd = [str(exe), "cora", "6", "2", "softmax", "16", "0", "0", "0.01", "2", "0", "3", "0"]
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GAIB_RANK", "GAIB_WORLD",
                                                                "GAIB_COMM", "GAIB_DEVICE", "GAIB_COMM_ID_FILE", "GAIB_REORDER")}
    base = dict(clean, DATASET_PATH=root, GAIB_COMM_TIMEOUT_S="60")
    if world > 1:
        base["GAIB_RANKS"] = str(world)
    plain = subprocess.run(cmd, capture_output=True, text=True, env=base, timeout=600)
    assert plain.returncode == 0, plain.stdout[-2000:] + plain.stderr[-2000:]
    re_run = subprocess.run(cmd, capture_output=True, text=True, env=dict(base, GAIB_REORDER=how), timeout=600)
    assert re_run.returncode == 0, re_run.stdout[-2000:] + re_run.stderr[-2000:]
    assert f"GAIB_REORDER={how}: vertices relabelled" in re_run.stdout
    pat = r"train_loss ([0-9.]+) train_acc ([0-9.]+)"
    want, got = re.findall(pat, plain.stdout), re.findall(pat, re_run.stdout)
    assert len(want) == 6 == len(got)
    for (wl, wa), (gl, ga) in zip(want, got):
        assert abs(float(wl) - float(gl)) <= 2e-3 and abs(float(wa) - float(ga)) <= 0.01, (want, got)
    pv = r"val_acc ([0-9.]+)"
    assert len(re.findall(pv, plain.stdout)) == len(re.findall(pv, re_run.stdout)) > 0
    for wa, ga in zip(re.findall(pv, plain.stdout), re.findall(pv, re_run.stdout)):
        assert abs(float(wa) - float(ga)) <= 0.01
    pt = r"Test accuracy: ([0-9.]+)"
    assert abs(float(re.search(pt, plain.stdout).group(1)) - float(re.search(pt, re_run.stdout).group(1))) <= 0.01
    bad = subprocess.run(cmd, capture_output=True, text=True, env=dict(base, GAIB_REORDER="metis"), timeout=120)
    assert bad.returncode != 0 and "cm, bfs or degree" in bad.stderr


def torch_device_count() -> int:
    import torch

    return torch.cuda.device_count()


def test_driver_rank_failure_exits_nonzero(tmp_path):
    """a rank whose peer never starts gives up after the deadline and exits non-zero (no hang)"""
    root, *_ = make_dataset(tmp_path)
    exe = ROOT / "bin" / "gpu_train_gcn"
    env = dict(os.environ, DATASET_PATH=root, RANK="0", WORLD_SIZE="2", GAIB_DEVICE="0", GAIB_COMM="ipc",
               GAIB_COMM_ID_FILE=str(tmp_path / "comm_id"), GAIB_COMM_TIMEOUT_S="3")
    r = subprocess.run([str(exe), "cora", "2", "1", "softmax"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "timed out" in r.stderr, r.stderr[-1500:]


@pytest.mark.parametrize("arch,loss", [("gcn", "softmax"), ("sage", "softmax"), ("gat", "softmax"), ("gcn", "sigmoid")])
def test_driver_recorded_epochs_equal_call_by_call(tmp_path, arch, loss):
    """GAIB_EPOCH_GRAPH: the epoch as two HIP-graph launches (forward + loss + metrics; backward + optimizer steps, the
    Adam beta powers advanced on the device) prints the same log as the epoch enqueued call by call"""
    root, x, labels, splits = make_dataset(tmp_path)
    exe = ROOT / "bin" / f"gpu_train_{arch}"
    cmd = [str(exe), "cora", "14", "2", loss, "16", "0", "0", "0.01", "2", "0", "5", "0"]
    out = {}
    for mode in ("0", "1"):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, DATASET_PATH=root, GAIB_EPOCH_GRAPH=mode))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert ("epochs recorded as HIP graphs" in r.stderr) == (mode == "1"), r.stderr[-1000:]
        out[mode] = (re.findall(r"train_loss ([0-9.]+) train_acc ([0-9.]+)", r.stdout),
                     re.findall(r"val_acc ([0-9.]+)", r.stdout), re.findall(r"Test accuracy: ([0-9.]+)", r.stdout))
    assert len(out["0"][0]) == 14 and len(out["0"][1]) == 2 and len(out["0"][2]) == 1
    assert out["0"] == out["1"]
    assert float(out["1"][0][-1][0]) < float(out["1"][0][0][0])  # it learns


def test_driver_recording_is_skipped_where_it_cannot_apply(tmp_path):
    """dropout draws its mask seed per call (a by-value argument): such a run stays call by call and says so"""
    root, *_ = make_dataset(tmp_path)
    cmd = [str(ROOT / "bin" / "gpu_train_gcn"), "cora", "4", "2", "softmax", "16", "0", "0.3", "0.01", "2", "0", "50", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, DATASET_PATH=root, GAIB_EPOCH_GRAPH="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GAIB_EPOCH_GRAPH=1 ignored" in r.stderr and "recorded as HIP graphs" not in r.stderr
