"""The world > 1 code on a GPU: two ranks share cuda:0 of the 1-GPU box, the collective runs over gloo
(RCCL refuses two ranks on one device), everything else is the production path: bench.py's N > 1 branch, and
PatchTrainer with the HIP ops and a real Comm, checked bit for bit against a host replay of the data-parallel rule."""
import json
import os
import random
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(script_and_args, timeout=900, nproc=2, **env):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_and_args
    e = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", **env)
    return subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_two_ranks_on_one_gpu():
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "8"],
                    ADV_BENCH_SHARE_GPU="1", ADV_BENCH_BACKEND="gloo")
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["pairs_per_gpu"] == 8
    assert d["value"] > 0 and abs(d["value"] - 2 * 8 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]     # whole-job pairs / max time
    assert d["roofline"]["clean_image_read_as"].startswith("uint8 index for 16 of 16")
    assert d["patch_allreduce"]["correct"] is True and d["patch_allreduce"]["avg_us"] > 0
    assert "cpu_baseline" not in d                                                                              # rank 0 at N = 1 only
    # what the ranks do end to end at N > 1 (beside `value`, never in it): every rank its own DSGN-shaped 20-step attack (image-sharded,
    # no collective) and a universal-patch epoch with the delta all-reduced inside the loop
    e = d["distributed_end_to_end"]
    assert "error" not in e, e
    a, p = e["image_sharded_attack"], e["universal_patch"]
    assert e["world"] == 2 and a["value"] > 0 and len(a["per_rank_s_per_attack"]) == 2 and a["loss_rose"] is True
    assert a["per_rank_pairs_per_s_min"] <= a["per_rank_pairs_per_s_max"] and abs(a["value"] - 2 * a["per_rank_pairs_per_s_min"]) < 1e-6 * a["value"]
    assert p["pairs"] == 4 and p["value"] > 0 and p["all_reduces_per_rank"] == 4 and p["message_bytes"] == 4 * (3 * 101 * 101 + 1)
    assert 0 < p["all_reduce_share_of_inner_iteration"] < 1 and len(p["all_reduce_ms_per_inner_iteration"]) == 2 and p["patch_abs_max"] > 0
    assert d["routes"]["mode"] == "table" and len(d["routes"]["hash"]) == 12 and d["routes"]["fixed_rule_lookups"] == 0


def test_bench_eight_ranks_on_one_gpu_under_ten_minutes():
    """the command the driver runs on an 8-GPU node, `... --nproc-per-node 8 bench.py --gpus 8 --steps K --warmup W`, with the eight ranks
    sharing cuda:0 (gloo for the collectives: RCCL refuses two ranks per device): one JSON line from rank 0, the process group says 8, one
    entry per rank, whole-job value = all pairs / the slowest rank's time - and the whole command well inside the driver's budget"""
    import time
    t0 = time.perf_counter()
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--pairs", "4", "--no-end-to-end"], nproc=8,
                    ADV_BENCH_SHARE_GPU="1", ADV_BENCH_BACKEND="gloo")
    took = time.perf_counter() - t0
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["rccl_world"] == 8 and d["backend"] == "gloo" and d["scaling"] == "weak" and d["config"]["pairs_per_gpu"] == 4
    assert len(d["per_rank"]["pairs_per_s"]) == 8 and len(d["per_rank"]["kernel_avg_launch_ms"]) == 8
    assert abs(d["value"] - 8 * d["per_rank"]["pairs_per_s_min"]) < 1e-6 * d["value"]
    assert d["config"]["parallelism"] == "image-sharded x8, no collective" and d["patch_allreduce"]["correct"] is True
    assert "cpu_baseline" not in d and "export_delivered" not in d                                      # rank 0 at N = 1 only
    assert took < 600, took


def test_bench_launches_its_own_ranks_when_started_plainly():
    """`python bench.py --gpus 2` with NO launcher around it (the way the driver starts --gpus 1): the parent starts the two ranks as a
    child torch.distributed.run, relays rank 0's one JSON line and its exit status; the line says what the process group says"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", ADV_BENCH_SHARE_GPU="1", ADV_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "8",
                          "--no-end-to-end"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_world"] == 2 and d["backend"] == "gloo" and d["launched_by"].startswith("bench.py itself")
    assert len(d["per_rank"]["pairs_per_s"]) == 2 and d["per_rank"]["pairs_per_s_min"] <= d["per_rank"]["pairs_per_s_max"]
    assert abs(d["value"] - 2 * d["per_rank"]["pairs_per_s_min"]) < 1e-6 * d["value"]          # whole job = all pairs / the slowest rank's time
    assert d["patch_allreduce"]["correct"] is True and d["patch_allreduce"]["avg_us"] > 0


@pytest.mark.parametrize("average", [False, True])
def test_patch_trainer_two_ranks_hip_ops_real_comm(tmp_path, average):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from oracle import oracle_np as O
    import synth
    n_pairs, H, W = 3, 384, 1248              # odd: rank 1 idles in the second round and contributes a zero delta
    out = _torchrun([os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), str(tmp_path), str(n_pairs), "1" if average else "0"])
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    p = [np.load(os.path.join(str(tmp_path), "patch_rank%d.npy" % k)) for k in range(2)]
    grads = [np.load(os.path.join(str(tmp_path), "grads_rank%d.npy" % k)) for k in range(2)]
    assert p[0].tobytes() == p[1].tobytes(), "ranks disagree on the patch"
    assert grads[0].shape[0] == 4 and grads[1].shape[0] == 2                      # 2 rounds x 2 iterations / 1 round x 2
    # host replay of the rule with the gradients the detector returned on each rank
    D, r = O.init_patch_dims(384, 0.2)
    patch = np.zeros((1, 3, D, D), np.float32)
    rngs = [random.Random(9 + 7919 * k) for k in range(2)]
    used = [0, 0]
    for rnd in range(2):
        live = []
        for rank in range(2):
            if rnd * 2 + rank < n_pairs:
                cl, cr = O.round_mask_centers(rngs[rank], H, W, r)
                live.append([rank, cl, cr, None])
        for it in range(2):
            total, count = None, 0
            for item in live:
                rank, cl, cr, gacc = item
                g = grads[rank][used[rank]]
                used[rank] += 1
                gacc = g if gacc is None else gacc + g
                item[3] = gacc
                d = O.patch_delta(gacc[0:1], gacc[1:2], cl[0], cl[1], cr[1], r, 8 / 255)
                total = d if total is None else total + d
                count += 1
            if average:
                total = total / np.float32(max(count, 1))
            patch = O.patch_apply_delta(patch, total)
    assert np.abs(patch).max() > 0
    assert p[0].reshape(patch.shape).tobytes() == patch.tobytes()
    assert os.path.exists(os.path.join(str(tmp_path), "dsgn_patch_ratio_0.2", "epoch1", "patch.npy"))


def test_bench_distributed_branch_over_rccl_with_one_rank():
    """the RCCL ("nccl") branch of bench.py - process group on the GPU, barrier(device_ids), MAX-reduce of the time, the
    patch-delta all-reduce probe - with a world of one rank (all a 1-GPU box can host; RCCL refuses two ranks per device)"""
    env = dict(os.environ, PYTHONPATH=ROOT, ADV_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs", "8",
                          "--no-cpu-baseline", "--no-end-to-end", "--no-srcnn", "--no-float-path"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["patch_allreduce"]["correct"] is True and "nccl" in d["patch_allreduce"]["collective"]


def test_attack_clis_under_torchrun_two_ranks(tmp_path):
    """the CLIs as `python -m torch.distributed.run --nproc-per-node 2 -m eval_driving_safety_amd.cli...`:
    PGD shards the pairs by image with no collective, the patch trainer all-reduces its delta and rank 0 writes the patch"""
    import numpy as np

    def launch(mod, argv):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), "-m", "eval_driving_safety_amd.cli." + mod] + argv
        e = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", ADV_COMM_BACKEND="gloo", ADV_SHARE_GPU="1")
        out = subprocess.run(cmd, cwd=str(tmp_path), env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-3000:]
        return out.stdout

    out = launch("dsgn_pgd_attack", ["--model", "toy", "--synthetic", "5", "-btest", "1", "--iter", "2", "--eps", "0.03"])
    assert "rank 0 attacked 3 stereo pairs" in out and "rank 1 attacked 2 stereo pairs" in out
    for k in range(3):
        assert sorted(os.listdir(str(tmp_path / ("dsgn_pgd_iters_%d" % k) / "image_2"))) == ["%06d.png" % i for i in range(5)]
    out = launch("dsgn_patch_attack", ["--model", "toy", "--synthetic", "3", "-btest", "1", "--iter", "1", "--epochs", "1", "--pos_seed", "4"])
    p = np.load(str(tmp_path / "dsgn_patch_ratio_0.2" / "epoch1" / "patch.npy"))
    assert p.shape == (1, 3, 77, 77) and np.abs(p).max() > 0 and out.count("Average loss for epoch1") == 1
