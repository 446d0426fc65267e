"""The `xflow_lr` binary and the local.sh-style launcher on a real GPU: the
reference's argv (main.cc:27-44), the reference's launch environment (scripts/local.sh:3-14:
DMLC_ROLE / DMLC_NUM_WORKER / DMLC_PS_ROOT_URI / DMLC_PS_ROOT_PORT).  Two workers share this
box's one GPU, so their exchange goes over the group's host transport (transport=host); with
one GPU per worker the same command line runs over RCCL."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as O
from xflow_amd import build, capi

from .test_gpu_parity import same
from .test_group_cpu import free_port

pytestmark = pytest.mark.gpu

CLI = os.path.join(build.LIBDIR, "xflow_lr")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_single_worker_prints_the_reference_metric_line(sample_prefixes, tmp_path):
    tr, te = sample_prefixes
    out = subprocess.run([CLI, tr, te, "0", "10", "capacity=4096",
                          "pred_path=" + str(tmp_path / "pred.txt")],
                         capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[0] == "start LR " and "my rank is = 0" in lines
    # the value SURVEY §4 records from the reference itself
    assert "logloss: -0.886206\tauc = 0.547149\ttp = 46 fp = 154" in lines
    assert lines[-2] == "train end......" and lines[-1].startswith("examples/sec")
    assert np.loadtxt(str(tmp_path / "pred.txt")).shape == (200, 3)


def test_cli_roles_without_a_process_return_at_once(sample_prefixes):
    tr, te = sample_prefixes
    for role in ("scheduler", "server"):
        out = subprocess.run([CLI, tr, te, "0", "1"], capture_output=True, text=True, timeout=60,
                             env=dict(os.environ, DMLC_ROLE=role))
        assert out.returncode == 0 and "no %s process" % role in out.stdout


@pytest.mark.parametrize("schedule", ["sequential", "owner", "owner ingest=gpu",
                                      "sequential ingest=gpu"])
def test_local_sh_two_workers_equal_the_rank_ordered_schedule(sample_prefixes, tmp_path, schedule):
    """(ingest=gpu: the workers' text tokenised on the GPU, xf_ingest.hip, and compiled from
    device arrays — the same tables and metric line.)
    scripts/local.sh 2 2 xflow_lr ...: worker r trains on small_train-0000r (identical files,
    SURVEY 2 row 18), both shards of the table take both workers' pushes in rank order, rank 0
    scores the test file against the whole table.  Checkpoint (one file per shard) and metric
    line against the oracle on that schedule."""
    tr, te = sample_prefixes
    ckpt = str(tmp_path / "model")
    env = dict(os.environ, DMLC_PS_ROOT_PORT=str(free_port()))
    out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "local.sh"), "2", "2", CLI, tr, te,
                          "0", "3", "transport=host", "capacity=256",
                          *("schedule=" + schedule).split(), "model_out=" + ckpt,
                          "pred_path=" + str(tmp_path / "pred.txt")],
                         capture_output=True, text=True, timeout=240, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout + out.stderr
    assert sorted(ln for ln in out.stdout.splitlines() if ln.startswith("my rank")) == \
        ["my rank is = 0", "my rank is = 1"]
    with O.sum_mode(1):
        s = O.Store(O.OPT_FTRL, 1)
        s.push(np.array([0], np.uint64), np.zeros(1, np.float32))     # lr_worker.cc:180-182
        for _ in range(3):
            blocks = [list(O.read_blocks("%s-%05d" % (tr, r), 2 << 20)) for r in range(2)]
            for b0, b1 in zip(*blocks):
                obs = [O.Batch(b[0], b[1], b[3]) for b in (b0, b1)]
                pulled = [s.pull(ob.ukeys) for ob in obs]
                grads = [ob.lr_grad(ob.lr_loss(pw)[0]) for ob, pw in zip(obs, pulled)]
                for ob, g in zip(obs, grads):
                    s.push(ob.ukeys, g)
        one = capi.Sharded(None, model="lr", optimizer="ftrl", capacity=4096)
        one.load(ckpt)                       # two shard files -> one table
        for a, e in zip(one.w.export(), s.export()):
            same(a, e)
        assert len(s) == 525                 # 524 train fids + key 0
        lab, p = O.predict(0, s, None, te + "-00000")
    ll, auc, tp, fp = O.auc_logloss(lab, p)
    assert O.format_auc_line(ll, auc, tp, fp) in out.stdout.splitlines()
    pred = np.loadtxt(str(tmp_path / "pred.txt"))
    assert pred.shape == (200, 3) and np.array_equal(pred[:, 2].astype(np.int32), lab)


def test_bench_runs_two_ranks_and_reports_them(tmp_path):
    """`python bench.py --gpus 2` spawns its own ranks (what the driver's scaling run does under
    torchrun): two ranks share this box's GPU over the host transport, a functional check of the
    whole N>1 bench path — group, C++ sharded trainer, stale1, the sequential profiling pass, the
    rank-0 JSON line."""
    import json
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--transport", "host", "--rows", "3000", "--nnz-per-row", "50",
                          "--keys-per-gpu", "200000", "--batches", "3", "--steps", "5",
                          "--warmup", "2"], capture_output=True, text=True, timeout=240,
                         env=dict(os.environ, MASTER_PORT=str(free_port()),
                                  XF_COLLECTIVE_TIMEOUT_S="60"))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["unit"] == "examples/sec"
    assert line["config"]["rows_per_gpu_batch"] == 3000
    assert len(line["config"]["shard_imbalance"]["owned_keys_per_step_by_rank"]) == 2
    assert set(line["kernels_ms"]) >= {"a2a_weights", "a2a_grads", "forward", "gradient"}
    assert abs(line["logloss"]["natural"] - 0.693) < 0.01
    # `value` comes from the owner-compute dataflow (its smoke run passed on both ranks), the
    # weight/gradient exchange ran as the supplementary leg
    assert "owner-compute" in line["config"]["parallelism"]
    ex = line["exchange_dataflow"]
    assert "error" not in ex and ex["value"] > 0
    st = line["owner_compute_sum_then_step"]
    assert "error" not in st and st["value"] > 0
    # FM over both ranks on the owner-compute dataflow (sum_then_step)
    fm = line["fm"]
    assert "error" not in fm and fm["value"] > 0 and "sum_then_step" in fm["dataflow"]


def test_one_worker_save_over_a_sharded_checkpoint_is_what_loads(sample_prefixes, tmp_path):
    """A sharded checkpoint (shard files + manifest) and then a single-table model saved under
    the SAME name: the later save is what a load must see (the stale manifest used to send the
    load to the old shard files)."""
    import numpy as np
    train, test = sample_prefixes
    path = str(tmp_path / "model")
    rng = np.random.default_rng(1)
    old = capi.Sharded(None, model="lr", optimizer="ftrl", capacity=1 << 12)
    ks = rng.choice(1 << 62, size=640).astype(np.uint64)
    old.step(old.compile(np.arange(65, dtype=np.uint64) * 10, ks,
                         rng.integers(0, 2, 64).astype(np.int32)))
    old.check()
    old.save(path)                                   # model.shard-00000-of-00001 + manifest
    assert os.path.exists(path + ".manifest")
    x = capi.XFlow(train, test, epochs=2)
    x.train()
    x.save(path)                                     # one worker: one file at `path`
    assert not os.path.exists(path + ".manifest")
    new = capi.Sharded(None, model="lr", optimizer="ftrl", capacity=1 << 12)
    new.load(path)
    assert len(new.w.export()[0]) == 877             # the sample data's keys, not the 640
    # ... and with a manifest that is older than the single file the single file still wins
    old.save(path + "2")
    x.save(path + "2")
    open(path + "2.manifest", "w").write("xflow_amd sharded model\nshards 1\nmodel 0\nk 0\n")
    os.utime(path + "2.manifest", (1, 1))
    newer = capi.Sharded(None, model="lr", optimizer="ftrl", capacity=1 << 12)
    newer.load(path + "2")
    assert len(newer.w.export()[0]) == 877
