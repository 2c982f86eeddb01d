"""Multi-process tile sharding on CPU (gloo, world_size 2): the host logic of raytracer_amd/dist.py -- tile ownership, the
tile-major layout, the gather and the assembly on rank 0 -- with the per-rank rendering done by the CPU oracle (test
infrastructure standing in for the GPU here).  The assembled frame must equal a single-process render bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from raytracer_amd import dist as rdist
from tests import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_assignment_covers_every_tile_once():
    for (w, h, world, ts) in [(1920, 1080, 8, 64), (250, 130, 3, 64), (64, 64, 2, 64), (3840, 2160, 4, 32)]:
        a = rdist.assign_tiles(w, h, world, ts)
        tx, ty = rdist.tile_grid(w, h, ts)
        assert sorted(np.concatenate(a).tolist()) == list(range(tx * ty))
        assert max(len(x) for x in a) - min(len(x) for x in a) <= 1


def test_tile_major_round_trip_with_partial_tiles():
    rng = np.random.default_rng(0)
    w, h, ts = 250, 130, 64
    img = rng.random((h, w, 3), np.float32)
    a = rdist.assign_tiles(w, h, 3, ts)
    bufs = [rdist.to_tile_major(img, t, ts) for t in a]
    for t, b in zip(a, bufs):
        assert b.shape == (len(t) * ts * ts, 3)
    back = rdist.assemble(bufs, a, w, h, ts)
    assert np.array_equal(back, img)
    # the first 64 entries of a tile are its first 8x8 pixel block, row-major
    ww, hh, valid = rdist.local_pixel_coords(a[0][:1], w, h, ts)
    assert list(ww[:9]) == [0, 1, 2, 3, 4, 5, 6, 7, 0] and list(hh[:9]) == [0] * 8 + [1] and ww[64] == 8 and hh[64] == 0


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = scenes.cube_scene()
    W, H, N, ts = 200, 136, 2, 64
    assignment = rdist.assign_tiles(W, H, world, ts)
    o = spec.build_oracle()
    tx, _ = rdist.tile_grid(W, H, ts)
    frame = np.zeros((H, W, 3), np.float32)
    for t in assignment[rank]:  # each rank renders only the tiles it owns
        x0, y0 = int(t % tx) * ts, int(t // tx) * ts
        r = o.render(spec.oracle_camera(), W, H, N, max_depth=12, window=(x0, y0, min(W, x0 + ts), min(H, y0 + ts)), workers=2)
        frame[y0:y0 + ts, x0:x0 + ts] = r["accumulator"][y0:y0 + ts, x0:x0 + ts]
    local = torch.from_numpy(rdist.to_tile_major(frame, assignment[rank], ts))
    bufs = rdist.gather_tiles(local, assignment, rank, world, dist)
    if rank == 0:
        full = rdist.assemble([b.numpy() for b in bufs], assignment, W, H, ts)
        np.save(out_path, full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_render_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frame.npy")
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    spec = scenes.cube_scene()
    want = spec.build_oracle().render(spec.oracle_camera(), 200, 136, 2, max_depth=12, workers=4)["accumulator"]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _dynamic_worker(rank, world, port, out_path):
    import time
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = scenes.cube_scene()
    W, H, N, ts = 200, 136, 2, 64
    tx, ty = rdist.tile_grid(W, H, ts)
    queue = rdist.TileQueue(rdist.make_queue_store(dist, rank, world), tx * ty, batch=2)
    o = spec.build_oracle()
    frame = np.zeros((H, W, 3), np.float32)
    mine = []
    while (batch := queue.next_batch()) is not None:
        for t in batch:
            x0, y0 = int(t % tx) * ts, int(t // tx) * ts
            r = o.render(spec.oracle_camera(), W, H, N, max_depth=12, window=(x0, y0, min(W, x0 + ts), min(H, y0 + ts)), workers=2)
            frame[y0:y0 + ts, x0:x0 + ts] = r["accumulator"][y0:y0 + ts, x0:x0 + ts]
        mine.extend(int(t) for t in batch)
        if rank == 1:
            time.sleep(0.3)  # a slow rank: the other one must end up with more tiles
    local = torch.from_numpy(rdist.to_tile_major(frame, np.asarray(mine, np.uint32), ts)) if mine else torch.zeros((0, 3))
    got = rdist.gather_dynamic(local, mine, rank, world, dist, ts * ts)
    if rank == 0:
        tiles = [g[0] for g in got]
        full = rdist.assemble([g[1].numpy() for g in got], tiles, W, H, ts)
        np.savez(out_path, frame=full, owner0=tiles[0], owner1=tiles[1])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dynamic_tile_queue_two_ranks_equals_single_process(tmp_path):
    """SURVEY 8d config 5's scheduling: tiles come from one shared cursor; every tile is rendered exactly once, the slower
    rank takes fewer, and the assembled frame is still bit-identical to a single-process render."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dyn.npz")
    mp.start_processes(_dynamic_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    z = np.load(out)
    tx, ty = rdist.tile_grid(200, 136, 64)
    assert sorted(z["owner0"].tolist() + z["owner1"].tolist()) == list(range(tx * ty))
    assert len(z["owner0"]) > len(z["owner1"])
    spec = scenes.cube_scene()
    want = spec.build_oracle().render(spec.oracle_camera(), 200, 136, 2, max_depth=12, workers=4)["accumulator"]
    assert np.array_equal(z["frame"].view(np.uint32), want.view(np.uint32))


def test_tile_queue_single_process_hands_out_every_tile_once():
    import torch.distributed as tdist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    store = tdist.TCPStore("127.0.0.1", port, 1, is_master=True, wait_for_workers=False)
    q = rdist.TileQueue(store, 11, 4)
    got = []
    while (b := q.next_batch()) is not None:
        got.append(b.tolist())
    assert got == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10]]


# ---- the self-launcher behind `python bench.py --gpus N` (raytracer_amd/dist.py launch_ranks)
_CHILD = """
import json, os, sys
keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RT_SELF_LAUNCHED", "HSA_ENABLE_IPC_MODE_LEGACY"]
rec = {k: os.environ.get(k) for k in keys}
open(os.path.join(sys.argv[1], "rank%s.json" % rec["RANK"]), "w").write(json.dumps(rec))
print("noise from rank", rec["RANK"])          # only rank 0's stdout may reach the caller
if rec["RANK"] == "0":
    print(json.dumps({"n_gpus": int(rec["WORLD_SIZE"])}))
sys.exit(int(sys.argv[2]) if rec["RANK"] == sys.argv[3] else 0)
"""


def test_launcher_starts_one_rank_per_gpu_with_the_distributed_environment(tmp_path):
    import json
    rc, text = rdist.launch_ranks([sys.executable, "-c", _CHILD, str(tmp_path), "0", "-1"], 3, backend="gloo", timeout=120)
    assert rc == 0
    lines = text.strip().splitlines()
    assert lines[0] == "noise from rank 0" and json.loads(lines[-1]) == {"n_gpus": 3} and len(lines) == 2
    recs = [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in range(3)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2"]
    assert {r["WORLD_SIZE"] for r in recs} == {"3"} and {r["MASTER_ADDR"] for r in recs} == {"127.0.0.1"}
    assert len({r["MASTER_PORT"] for r in recs}) == 1 and int(recs[0]["MASTER_PORT"]) > 0
    assert {r["HSA_ENABLE_IPC_MODE_LEGACY"] for r in recs} == {"0"}


def test_launcher_fails_when_a_rank_fails_and_refuses_more_ranks_than_gpus(tmp_path):
    rc, _ = rdist.launch_ranks([sys.executable, "-c", _CHILD, str(tmp_path), "7", "1"], 2, backend="gloo", timeout=120)
    assert rc == 7
    with pytest.raises(RuntimeError, match="only 1 GPU"):
        rdist.launch_ranks([sys.executable, "-c", "pass"], 2, backend="nccl", devices=1)


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` on a box with fewer GPUs must say so, not run on one GPU and print n_gpus: 1."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RT_DIST_BACKEND")}
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("more GPUs than the test asks for")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "GPU(s) are visible" in p.stderr and p.stdout.strip() == ""
