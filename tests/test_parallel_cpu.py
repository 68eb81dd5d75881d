"""Prompt sharding + weight broadcast with world_size 2 on CPU (gloo), and the dataset parsing rules."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from oracle import golden_inputs as gi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(REPO, "diffusion-spacetime-attn_amd"))
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from sta import parallel, synth
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                              # different garbage on every rank before the broadcast
    def make():    # Linear + norm + a channels_last (NHWC-strided) conv weight, as in the NHWC UNet trunk + a buffer
        m = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.LayerNorm(96), torch.nn.Linear(96, 8),
                                torch.nn.Conv2d(16, 24, 3), torch.nn.Linear(129, 131)).to(torch.bfloat16).to(memory_format=torch.channels_last)
        m.register_buffer("table", torch.randn(33))
        return m
    net = make()
    if rank == 0:
        synth.seeded_fill_(net, 7)
    # small buckets: several messages per dtype; the 129 x 131 Linear is one bucket of odd length that goes as
    # scatter + all-gather with padding, the small ones as plain broadcasts
    nbytes = parallel.broadcast_module_(net, src=0, bucket_bytes=4096)
    ref = make()
    synth.seeded_fill_(ref, 7)
    same = all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))
    same = same and net[3].weight.is_contiguous(memory_format=torch.channels_last)      # the layout survives the broadcast
    mine = parallel.shard_indices(7, rank, world)
    t = parallel.max_over_ranks(float(rank + 1), torch.device("cpu"))
    parallel.barrier()
    json.dump({"same": same, "mine": mine, "nbytes": nbytes, "max": t}, open(os.path.join(out_dir, "r%d.json" % rank), "w"))


def test_broadcast_and_sharding_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [json.load(open(tmp_path / ("r%d.json" % r))) for r in range(2)]
    assert all(r["same"] for r in res)                          # every rank holds rank 0's weights, buffers included
    assert res[0]["mine"] == [0, 2, 4, 6] and res[1]["mine"] == [1, 3, 5]
    assert sorted(res[0]["mine"] + res[1]["mine"]) == list(range(7))
    assert res[0]["nbytes"] == res[1]["nbytes"] > 0
    assert res[0]["max"] == res[1]["max"] == 2.0


def test_single_process_is_a_no_op():
    from sta import parallel
    net = torch.nn.Linear(4, 4)
    assert parallel.broadcast_module_(net) == 0
    assert parallel.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert parallel.max_over_ranks(3.5, torch.device("cpu")) == 3.5


def test_dataset_parsing_rules():
    from sta import datasets
    g = json.load(open(os.path.join(gi.GOLDEN, "prompts.json")))
    assert datasets.parse_prompts("\n".join(g["gpt_head"]), "gpt", limit=4) == g["gpt_prompts"]
    for kind in ("mscoco", "vsr"):
        assert datasets.parse_prompts("\n".join(g[kind + "_head"]), kind, limit=4) == g[kind + "_prompts"]
    with pytest.raises(ValueError):
        datasets.parse_prompts("x", "imagenet")
    lay = {"a cat": {"cat": [0.5, 0.5]}, "3": {"dog": [0.1, 0.2]}}
    assert datasets.layout_for(lay, "a cat", 0) == {"cat": [0.5, 0.5]}
    assert datasets.layout_for(lay, "zzz", 3) == {"dog": [0.1, 0.2]}
    assert datasets.layout_for(lay, "zzz", 9) is None and datasets.layout_for(None, "a cat", 0) is None


def test_crop_box_rule():
    from ldm.models.diffusion.plms import object_crop_box
    y1, y2, x1, x2 = object_crop_box((0.3, 0.4), 512, 512)
    assert (y1, y2, x1, x2) == (int(512 * max(0.4 - 0.2, 0)), int(512 * min(0.4 + 0.2, 1)), int(512 * max(0.3 - 0.2, 0)), int(512 * min(0.3 + 0.2, 1)))
    assert object_crop_box((0.05, 0.95), 512, 512) == (384, 512, 0, 128)      # clipped to the image (plms.py:262-265)


def test_entry_point_cli_surface():
    sys.path.insert(0, os.path.join(REPO, "diffusion-spacetime-attn_amd", "scripts"))
    import _txt2img_common as c
    p = c.build_parser("x.txt")
    a = p.parse_args(["--plms", "--ddim_steps", "50", "--prompt", "", "--scale", "7.5", "--H", "512", "--W", "512",
                      "--n_samples", "1", "--seed", "42", "--process_id", "3", "--precision", "autocast", "--fixed_code"])
    assert a.plms and a.ddim_steps == 50 and a.C == 4 and a.f == 8 and a.opt_epochs == 3 and a.dataset == "x.txt"


def _run_bench(*flags):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *flags], env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_bench_starts_its_own_ranks_dry_launch():
    """`python bench.py --gpus 2` (the way the driver calls it: no torch.distributed environment) starts two ranks itself; with
    --dry-launch they form the group (gloo here, RCCL on GPUs), move rank 0's weights to rank 1 in scatter + all-gather buckets,
    verify them and stop before the first sampler kernel. ONE JSON line, rc 0."""
    r, lines = _run_bench("--gpus", "2", "--dry-launch", "--images-per-step", "32")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["dry_launch"] and out["n_gpus"] == 2 and out["backend"] == "gloo"
    assert out["weights_identical_on_every_rank"] and out["weight_broadcast_bytes"] > 0
    # 2 x 32 prompts = the 64-prompt set of BASELINE configs[3] in disjoint slices (the default, 64 per rank, walks the set once per rank)
    assert out["global_batch"] == 64 and out["first_prompts_of_step0"] == [0, 1, 32, 33]


def test_bench_eight_ranks_dry_launch_is_the_config3_split():
    """`python bench.py --gpus 8 --scaling strong --dry-launch` — BASELINE configs[3] as the driver's 8-GPU run will start it, minus the
    kernels: eight self-started ranks over gloo, one rendezvous on 127.0.0.1, rank 0's weights on every rank, and per rank its own eight
    prompts of the 64 (disjoint, together all of them), its own local rank / device slot and its own MIOpen user-db copy."""
    r, lines = _run_bench("--gpus", "8", "--dry-launch", "--scaling", "strong")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["backend"] == "gloo" and out["weights_identical_on_every_rank"]
    assert out["scaling"] == "strong" and out["images_per_step"] == 8 and out["global_batch"] == 64
    ranks = sorted(out["ranks"], key=lambda x: x["rank"])
    assert [x["rank"] for x in ranks] == list(range(8)) and [x["local_rank"] for x in ranks] == list(range(8))
    assert all(len(x["prompts_step0"]) == 8 for x in ranks)
    assert sorted(p for x in ranks for p in x["prompts_step0"]) == list(range(64))            # disjoint and complete
    assert len({x["master"] for x in ranks}) == 1 and ranks[0]["master"].startswith("127.0.0.1:")
    assert len({x["pid"] for x in ranks}) == 8
    dbs = [x["miopen_user_db"] for x in ranks]
    assert all(d is None for d in dbs) or (len(set(dbs)) == 8 and all(d.endswith("rank%d" % i) for i, d in enumerate(dbs)))


def test_bench_strong_scaling_split_and_world_mismatch():
    r, lines = _run_bench("--gpus", "2", "--dry-launch", "--scaling", "strong")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["scaling"] == "strong" and out["images_per_step"] == 32 and out["global_batch"] == 64      # 64 prompts / 2 ranks
    # a torch.distributed environment that disagrees with --gpus is refused (non-zero rc, no JSON line)
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]
