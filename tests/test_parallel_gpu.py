"""The RCCL (`backend="nccl"`) code path of sta.parallel on the one GPU a test box has.

8-GPU runs are the driver's; what can be executed here is (a) process-group initialisation + the collectives of
broadcast_module_ / max_over_ranks through RCCL in a 1-rank group, and (b) a 2-rank group whose ranks share GPU 0 —
RCCL refuses duplicate devices in one communicator on some builds; the test then records the refusal (skip with the
reason) instead of pretending. The multi-rank logic itself (sharding, scatter + all-gather buckets, layouts) is
covered with world_size 2 on gloo in tests/test_parallel_cpu.py."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(REPO, "diffusion-spacetime-attn_amd"))
    dev_id = rank if torch.cuda.device_count() >= world else 0          # one GPU per rank when the box has them
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(dev_id), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from sta import parallel, synth
    res = {"ok": False}
    try:
        torch.cuda.set_device(dev_id)
        if world == 1:
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        else:
            parallel.init_from_env(backend="nccl")
        net = torch.nn.Sequential(torch.nn.Linear(256, 384), torch.nn.LayerNorm(384), torch.nn.Conv2d(16, 24, 3)).to(torch.float16).cuda()
        if rank == 0:
            synth.device_fill_(net, 7)
        ref = torch.nn.Sequential(torch.nn.Linear(256, 384), torch.nn.LayerNorm(384), torch.nn.Conv2d(16, 24, 3)).to(torch.float16).cuda()
        synth.device_fill_(ref, 7)
        if world == 1:       # broadcast_module_ returns early for one rank: drive the same collectives directly through RCCL
            flat = torch.cat([t.reshape(-1) for t in net.state_dict().values() if t.dtype == torch.float16])
            before = flat.clone()
            parallel._broadcast_flat(flat, 0, 1)
            same = torch.equal(flat, before)
            t = torch.tensor([3.5], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res.update(max=float(t.item()), nbytes=flat.numel() * 2)
        else:
            nbytes = parallel.broadcast_module_(net, src=0, bucket_bytes=64 << 10)
            same = all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))
            res.update(max=parallel.max_over_ranks(float(rank + 1), torch.device("cuda", dev_id)), nbytes=nbytes)
            parallel.barrier()
        torch.cuda.synchronize()
        res.update(ok=bool(same), backend=dist.get_backend())
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001 — the refusal text is the result
        res["error"] = repr(e)[:400]
    json.dump(res, open(os.path.join(out_dir, "r%d.json" % rank), "w"))


def test_rccl_collectives_one_rank(tmp_path):
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = json.load(open(tmp_path / "r0.json"))
    assert r.get("ok") and r["backend"] == "nccl" and r["max"] == 3.5 and r["nbytes"] > 0, r


def test_rccl_broadcast_two_ranks(tmp_path):
    """The real N > 1 path (scatter + all-gather buckets, max over ranks) through RCCL: one rank per GPU when the box has two
    or more GPUs. On a one-GPU box two ranks would have to share the device — outside what RCCL supports — so the test
    is skipped there unless STA_TEST_RCCL_2RANK=1 asks for the attempt."""
    if torch.cuda.device_count() < 2 and os.environ.get("STA_TEST_RCCL_2RANK") != "1":
        pytest.skip("this box has %d GPU(s): the 2-rank RCCL broadcast needs one rank per GPU (STA_TEST_RCCL_2RANK=1 tries two ranks on one "
                    "device); the N > 1 plumbing itself runs in test_bench_two_ranks_sharing_one_gpu_end_to_end and, without kernels, on 8 "
                    "gloo ranks in tests/test_parallel_cpu.py" % torch.cuda.device_count())
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    res = [json.load(open(tmp_path / ("r%d.json" % r))) for r in range(2)]
    if torch.cuda.device_count() < 2 and any("error" in r for r in res):
        pytest.skip("RCCL does not form a 2-rank communicator on one GPU here: %s" % [r.get("error") for r in res])
    assert all(r["ok"] for r in res), res
    assert all(r["ok"] for r in res) and res[0]["nbytes"] == res[1]["nbytes"] > 0 and res[0]["max"] == res[1]["max"] == 2.0


def _bench(*flags, timeout=900):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *flags], env=env, capture_output=True, text=True, timeout=timeout)
    return r, [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_dry_launch_on_the_gpu():
    """`python bench.py --gpus N --dry-launch` with the REAL SD-v1 weights: N = 2 over RCCL when the box has two GPUs (bench.py
    starts its own ranks), N = 1 otherwise (the same code path minus the collectives)."""
    n = 2 if torch.cuda.device_count() >= 2 else 1
    r, lines = _bench("--gpus", str(n), "--dry-launch")
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["dry_launch"] and out["n_gpus"] == n and out["device"] == "cuda" and out["weights_identical_on_every_rank"]
    if n == 2:
        assert out["backend"] == "nccl" and out["weight_broadcast_bytes"] > 1.5e9      # UNet + VAE in 16 bit


def test_bench_two_ranks_end_to_end():
    """The whole N = 2 bench (weights over RCCL, per-rank MIOpen db copies, hipGraph capture in two processes, barrier + max over
    ranks) at a small size; needs two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("this box has %d GPU(s): the 2-rank end-to-end bench over RCCL needs one rank per GPU; --share-gpu (next test) runs the "
                    "same path with both ranks on GPU 0 and gloo collectives" % torch.cuda.device_count())
    r, lines = _bench("--gpus", "2", "--steps", "1", "--warmup", "1", "--images-per-step", "2", "--ddim_steps", "4", timeout=1500)
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["weight_broadcast_bytes"] > 1.5e9


def test_bench_two_ranks_sharing_one_gpu_end_to_end():
    """What a one-GPU box CAN execute of the N = 2 bench: `--share-gpu` puts both ranks on GPU 0 and lets gloo carry
    the collectives (RCCL refuses duplicate devices); everything else is the real multi-rank path — bench.py starting its own ranks,
    rank 0's weights reaching rank 1 through the bucketed scatter + all-gather (1.9 GB, host-staged here), a MIOpen user-db copy per rank,
    the CFG UNet call captured into a hipGraph in two processes at once, disjoint prompt shards, barrier + max over ranks, ONE JSON line."""
    r, lines = _bench("--gpus", "2", "--share-gpu", "--steps", "1", "--warmup", "1", "--images-per-step", "2", "--ddim_steps", "4",
                      "--no-cpu-baseline", "--no-side-runs", "--no-roofline", timeout=1500)
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["global_batch"] == 4
    assert out["config"]["weight_broadcast_bytes"] > 1.5e9 and "TEST MODE" in out["config"]["parallelism"]
    r, lines = _bench("--gpus", "2", "--share-gpu", "--dry-launch")
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["weights_identical_on_every_rank"] and d["backend"] == "gloo" and d["device"] == "cuda" and d["n_gpus"] == 2


def test_bench_eight_ranks_sharing_one_gpu_strong_scaling_split():
    """The driver's N = 8 launch as far as one GPU can execute it: eight ranks on GPU 0 (gloo carries the collectives), BASELINE configs[3]'s
    64 prompts split 8 x 8 (`--scaling strong`), weights from rank 0 to seven peers, eight MIOpen user-db copies, eight hipGraph captures at
    once, barrier + max over ranks, ONE JSON line. Reduced schedule (4 PLMS steps); the full 50-step run of the same command is
    profiles/r06_share8.json."""
    r, lines = _bench("--gpus", "8", "--share-gpu", "--scaling", "strong", "--steps", "1", "--warmup", "1", "--ddim_steps", "4",
                      "--no-cpu-baseline", "--no-side-runs", "--no-roofline", timeout=1500)
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["config"]["global_batch"] == 64 and out["config"]["images_per_step"] == 8
    assert out["config"]["weight_broadcast_bytes"] > 1.5e9 and "TEST MODE" in out["config"]["parallelism"]

