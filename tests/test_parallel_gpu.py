"""Data parallelism on the HIP path, executed on the GPU box (one MI355X): fresh child processes (never a re-exec of the pytest
process) share cuda:0, sharded 1 + 1 over gloo, and must reproduce the single-process global-batch run -- for the benchmarked
`DataParallelStep` and for the reference-surface `Trainer.train` loop (reference: main.py:61, train.py:201-223).
The RCCL transport itself (`backend="nccl"`) is executed with one rank (two ranks on one device are refused by RCCL)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_gpu_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(mode, out_dir, world, backend="gloo"):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, str(out_dir), backend], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode("utf-8", "replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-4000:])
    return [dict(np.load(os.path.join(str(out_dir), "%s_w%d_r%d.npz" % (mode, world, r)))) for r in range(world)]


def _rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))


@pytest.mark.timeout(1800)
def test_two_ranks_on_one_gpu_equal_the_global_batch_step(tmp_path):
    (ref,) = _launch("step", tmp_path, 1)
    r0, r1 = _launch("step", tmp_path, 2)
    # the ranks agree bit for bit after the all-reduces (same reduction result on both, same Adam)
    assert r0["loss"] == r1["loss"] and np.array_equal(r0["grads"], r1["grads"]) and np.array_equal(r0["weights"], r1["weights"])
    # ... and reproduce the single-process batch-2 step: only summation order differs (per-sample partials + all-reduce vs one pass)
    assert abs(float(r0["loss"]) - float(ref["loss"])) < 2e-6 and abs(float(r0["dice"]) - float(ref["dice"])) < 2e-6
    assert abs(float(r0["l_step1"]) - float(ref["l_step1"])) < 2e-6 and abs(float(r0["l_step2"]) - float(ref["l_step2"])) < 1e-4
    rel = _rel(r0["grads"], ref["grads"])
    print("sharded vs global-batch gradient bucket: relative L2 %.2e" % rel)
    assert rel < 2e-3, rel                                  # missing all-reduce: ~0.7; naive DDP (per-rank Dice, mean): ~6e-3
    # Adam normalises: |dw| <= lr per step whatever the gradient magnitude; two steps at lr 1e-3, 5e-4
    dw = np.abs(r0["weights"] - ref["weights"])
    assert dw.max() <= 2 * (1e-3 + 5e-4) + 1e-6 and dw.mean() < 2e-5, (dw.max(), dw.mean())
    live = r0["grads"] != 0
    assert 0.6 < live.mean() < 0.9                          # the dead third of the bucket stayed zero on both


@pytest.mark.timeout(1800)
def test_trainer_loop_is_data_parallel_on_the_hip_path(tmp_path):
    """the reference's own loop (Trainer.train) under a 2-rank process group: per-rank shards, global-batch criteria, summed
    gradients, identical replicas, rank 0 writes the checkpoints -- equal to the single-process run on the global batches"""
    (ref,) = _launch("trainer", tmp_path, 1)
    r0, r1 = _launch("trainer", tmp_path, 2)
    assert np.array_equal(r0["weights"], r1["weights"])     # replicas stay identical
    assert int(r0["global_step"]) == int(ref["global_step"]) == 2
    np.testing.assert_allclose(r0["losses"], ref["losses"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(r0["losses"], r1["losses"], rtol=0, atol=0)
    dw = np.abs(r0["weights"] - ref["weights"])
    assert dw.max() <= 2 * (1e-3 + 5e-4) + 1e-6 and dw.mean() < 2e-5, (dw.max(), dw.mean())
    np.testing.assert_allclose(r0["train_dice"], ref["train_dice"], atol=2e-3)      # global-batch metric from the shard metrics
    np.testing.assert_allclose(r0["train_dice"], r1["train_dice"], atol=0)
    np.testing.assert_allclose(r0["val_dice"], ref["val_dice"], atol=2e-3)
    assert int(r0["ckpt"]) == 1 and int(ref["ckpt"]) == 1


@pytest.mark.timeout(900)
def test_rccl_transport_executes_with_one_rank(tmp_path):
    """backend "nccl" IS RCCL on ROCm: init_process_group + the step's all-reduces (criterion sums, live gradient runs) + barrier run
    through librccl on the GPU; with one rank the result must equal the run without a process group."""
    (ref,) = _launch("step", tmp_path, 1)
    ref = {k: v.copy() for k, v in ref.items()}
    (got,) = _launch("step", tmp_path, 1, backend="nccl")
    assert float(got["loss"]) == float(ref["loss"]) and np.array_equal(got["grads"], ref["grads"]) and np.array_equal(got["weights"], ref["weights"])


@pytest.mark.timeout(900)
def test_rccl_data_group_on_a_gloo_default_group_executes_with_one_rank(tmp_path):
    """The process-group set-up bench.py's ranks use under the driver's launcher (parallel.init_process_groups_with_fallback: gloo default group,
    RCCL data group created on top, health-check all-reduce, agreement over gloo) executed on the HEALTHY path with one rank: the data group is
    RCCL, no fallback reason, and the step through `process_group=<that group>` equals the run without any process group bit for bit."""
    (ref,) = _launch("step", tmp_path, 1)
    ref = {k: v.copy() for k, v in ref.items()}
    (got,) = _launch("step", tmp_path, 1, backend="gloo+nccl")
    assert float(got["loss"]) == float(ref["loss"]) and np.array_equal(got["grads"], ref["grads"]) and np.array_equal(got["weights"], ref["weights"])


@pytest.mark.timeout(900)
def test_library_rccl_entry_points_execute_with_one_rank(tmp_path):
    """ru_comm_unique_id / ru_comm_init / ru_allreduce (include/resunet_hip.h): the step's collectives through the library's own RCCL
    binding on the kernels' stream, no torch.distributed call in the data path.  One rank (RCCL refuses two ranks on one device):
    the all-reduce is the identity and the step must equal the run without any communicator."""
    (ref,) = _launch("step", tmp_path, 1)
    ref = {k: v.copy() for k, v in ref.items()}
    (got,) = _launch("step", tmp_path, 1, backend="rccl-direct")
    assert int(got["comm_world"]) == 1
    np.testing.assert_array_equal(got["probe32"], np.arange(1000, dtype=np.float32) * 0.5)
    np.testing.assert_array_equal(got["probe64"], np.arange(7, dtype=np.float64) + 0.25)
    assert float(got["loss"]) == float(ref["loss"]) and np.array_equal(got["grads"], ref["grads"]) and np.array_equal(got["weights"], ref["weights"])


@pytest.mark.timeout(1800)
def test_bare_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` started WITHOUT a launcher (how the driver starts the scaling runs) spawns its two ranks through
    torch.distributed.run as fresh child processes and relays rank 0's single JSON line: n_gpus 2, weak scaling, whole-job `value`, and
    the forward leg (2 replicas) that BASELINE's metric asks for at every N.  One MI355X here: the ranks share cuda:0 over gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--size", "64", "--batch", "1", "--no-cpu-baseline", "--no-power", "--probe-steps", "0"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and abs(out["value"] - 2 * 1e3 / out["ms_per_step"]) < 1e-2 * out["value"]
    assert out["fwd"]["replicas"] == 2 and out["fwd"]["value"] > 0
    assert out["fwd_batch"]["replicas"] == 2 and out["fwd_batch"]["value"] > 0
    assert np.isfinite(out["final_loss"])
    # what explains a scaling curve from the line alone (round 4): the collectives timed in place, per-rank step times, who talked to whom
    ar = out["allreduce_ms"]
    assert ar["gradients"] > 0 and ar["criterion_sums"] > 0 and len(ar["per_rank_gradients"]) == 2 and ar["gradient_bytes"] > 4 * 4_000_000
    pr = out["step_ms_per_rank"]
    assert len(pr["all"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - out["ms_per_step"]) < 1e-2 * out["ms_per_step"] + 1e-3
    assert out["ranks"] == 2 and out["rccl_ranks"] == 0 and out["dist_backend"] == "gloo" and "gloo" in out["transport"] and "degraded" not in out


@pytest.mark.timeout(2400)
@pytest.mark.parametrize("mode", ["health-check-injected", "health-check-real", "ranks-die"])
def test_bench_prints_a_flagged_gloo_line_when_rccl_does_not_come_up(mode):
    """A degraded line instead of no line (round-4 verdict, item 7), on the HIP path: (1) the in-rank route the DRIVER's launcher form takes --
    parallel.init_process_groups_with_fallback: RCCL health check fails (injected; and for real: two ranks on ONE device, which RCCL
    refuses), the ranks agree over gloo and keep going; (2) the bare launcher's route -- the first set of ranks dies over RCCL before any GPU
    call, ONE fresh gloo set is started.  Either way: exit code 0, one JSON line, `transport` = "gloo (fallback after RCCL failure: ...)",
    `degraded`, `rccl_ranks` 0, and a real measurement behind `value`."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["RU_BENCH_INJECT_RCCL_FAIL"] = {"health-check-injected": "1", "health-check-real": "", "ranks-die": "2"}[mode]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--steps", "2", "--warmup", "1", "--size", "32", "--batch", "1",
                        "--no-extras", "--probe-steps", "0"], env=env, capture_output=True, text=True, timeout=2000)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["rccl_ranks"] == 0 and out["dist_backend"] == "gloo" and out["degraded"] is True
    assert out["transport"].startswith("gloo (fallback after RCCL failure: "), out["transport"]
    if mode != "health-check-real":
        assert "injected RCCL failure" in out["transport"]
    else:
        print("real RCCL failure with two ranks on one device:", out["transport"])
    assert out["value"] > 0 and np.isfinite(out["final_loss"]) and out["allreduce_ms"]["gradients"] > 0
    assert (r.stderr.count("starting ONE fresh set of ranks") == 1) == (mode == "ranks-die")
