"""Multi-GPU tile sharding: one process per GPU, full scene replica per rank, image tiles as the data-parallel unit,
one gather of the fp32 accumulator at the end (RCCL over xGMI when the backend is `nccl`, gloo in the CPU tests).

A path depends only on (w, h, samp), the camera and the scene (src/cpu/CPURenderer.cpp:156-198), so ranks never
exchange anything while rendering; per-pixel accumulation over `samp` stays inside one rank, which keeps the
assembled image bit-identical to a single-GPU render.
"""
import os
import socket
import subprocess
import sys

import numpy as np


# ---------------------------------------------------------------------------------------------------------------------
# Self-launch: `python bench.py --gpus N` with no launcher around it (the reference is one process, src/main.cpp:6-56; here
# one process per GPU).  The parent touches no GPU: it only starts N fresh children with the torch.distributed environment
# (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), waits for them, relays rank 0's stdout and fails if any failed.

def free_port(host="127.0.0.1"):
    with socket.socket() as s:
        s.bind((host, 0))
        return s.getsockname()[1]


def rank_env(rank, world_size, port, base=None, host="127.0.0.1"):
    """Environment of rank `rank` of a one-node job of `world_size` processes (what torch.distributed.run would set)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), LOCAL_WORLD_SIZE=str(world_size),
               MASTER_ADDR=host, MASTER_PORT=str(port), RT_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    return env


def visible_gpus():
    """Number of GPUs this process may use, counted without initialising one (the launcher parent must stay GPU-free)."""
    import torch
    return int(torch.cuda.device_count())


def launch_ranks(argv, world_size, backend="nccl", devices=None, timeout=None):
    """Starts `argv` (a command line, e.g. [sys.executable, 'bench.py', ...]) once per rank, waits for all of them and returns
    (exit code, rank 0's stdout).  A rank that fails takes the others down (they would wait in a collective for ever).
    With `nccl` every rank needs a GPU of its own: more ranks than visible devices is an error, not a silent single-GPU
    run; `gloo` rehearsals may share devices."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    if backend == "nccl":
        have = visible_gpus() if devices is None else int(devices)
        if world_size > have:
            raise RuntimeError("--gpus %d asked for but only %d GPU(s) are visible: one rank per GPU over RCCL "
                               "(RT_DIST_BACKEND=gloo rehearses N ranks on fewer devices)" % (world_size, have))
    port = free_port()
    procs = []
    for rank in range(world_size):
        out = subprocess.PIPE if rank == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen(list(argv), env=rank_env(rank, world_size, port), stdout=out, stderr=None))
    failed = 0
    text = ""
    try:
        # rank 0 is the only one that prints; reading it to the end also waits for it
        text = procs[0].communicate(timeout=timeout)[0].decode("utf-8", "replace")
        for p in procs:
            rc = p.wait(timeout=timeout)
            failed = failed or rc
    except subprocess.TimeoutExpired:
        failed = 124
    finally:
        for p in procs:  # exactly the children started above, by handle
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return (failed if failed else 0), text


def tile_grid(width, height, tile_size=64):
    return (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size


def assign_tiles(width, height, world_size, tile_size=64):
    """Static round-robin ownership: rank r owns tiles r, r+world, ... of the row-major tile list.  Neighbouring tiles go
    to different GPUs, so every rank sees a similar mix of cheap and expensive image regions."""
    tx, ty = tile_grid(width, height, tile_size)
    ids = np.arange(tx * ty, dtype=np.uint32)
    return [balanced_order(ids[r::world_size], tx) for r in range(world_size)]


def balanced_order(tile_ids, tiles_x):
    """A rank's tiles with those of the even tile rows first, then those of the odd rows (row-major inside each group).  The
    kernel library cuts a frame's tile list into contiguous batches; in plain row-major order a two-batch frame is the upper and
    the lower part of the image, which cost very different amounts under a horizon -- interleaved rows halve any image evenly
    (the library's own default list is ordered the same way, raytracer_amd/csrc/rthip.hip rthip_begin)."""
    t = np.asarray(tile_ids, dtype=np.uint32)
    odd = ((t // np.uint32(tiles_x)) & np.uint32(1)).astype(bool)
    return np.ascontiguousarray(np.concatenate([t[~odd], t[odd]]), dtype=np.uint32)


def padded_tile_count(assignment):
    return max(len(a) for a in assignment)


def local_pixel_coords(tile_ids, width, height, tile_size=64):
    """(w, h, valid) of every pixel of a rank's tile-major buffer, in the device's order: tiles in list order, 8x8 pixel
    blocks row-major inside a tile, pixels row-major inside a block (raytracer_amd/csrc/rthip.hip pixelOf)."""
    tx, _ = tile_grid(width, height, tile_size)
    tile_ids = np.asarray(tile_ids, dtype=np.int64)
    per = tile_size * tile_size
    k = np.arange(per, dtype=np.int64)
    blk, lane = k >> 6, k & 63
    bpr = tile_size >> 3
    px = (blk % bpr) * 8 + (lane & 7)
    py = (blk // bpr) * 8 + (lane >> 3)
    w = (tile_ids % tx)[:, None] * tile_size + px[None, :]
    h = (tile_ids // tx)[:, None] * tile_size + py[None, :]
    valid = (w < width) & (h < height)
    return w.reshape(-1), h.reshape(-1), valid.reshape(-1)


def to_tile_major(image, tile_ids, tile_size=64):
    """Row-major (H, W, 3) -> the tile-major (numTiles*ts*ts, 3) buffer a rank owning `tile_ids` would hold."""
    hgt, wid = image.shape[:2]
    w, h, valid = local_pixel_coords(tile_ids, wid, hgt, tile_size)
    out = np.zeros((w.size, 3), image.dtype)
    out[valid] = image[h[valid], w[valid]]
    return out


def assemble(gathered, assignment, width, height, tile_size=64):
    """Rank 0: scatter every rank's tile-major buffer into one row-major (H, W, 3) frame (numpy reference of
    rthip_untile)."""
    frame = np.zeros((height, width, 3), np.float32)
    for buf, tiles in zip(gathered, assignment):
        w, h, valid = local_pixel_coords(tiles, width, height, tile_size)
        buf = np.asarray(buf).reshape(-1, 3)[: w.size]
        frame[h[valid], w[valid]] = buf[valid]
    return frame


def gather_tiles(local, assignment, rank, world_size, dist, device=None):
    """Gathers equal-sized (padded) tile-major tensors to rank 0.  `local` is a torch tensor (numLocalPixels, 3)."""
    import torch
    per_rank = padded_tile_count(assignment) * (local.shape[0] // max(1, len(assignment[rank])))
    if local.shape[0] < per_rank:
        pad = torch.zeros((per_rank - local.shape[0], 3), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    local = local.contiguous()
    if world_size == 1:
        return [local]
    bufs = [torch.empty_like(local) for _ in range(world_size)] if rank == 0 else None
    dist.gather(local, bufs, dst=0)
    return bufs


def init_native_comm(renderer, dist, rank, world_size):
    """Joins the renderer's context to an RCCL communicator of its own (include/rthip.h: rthip_comm_init) so that
    rthip_gather can assemble the frame inside the C ABI.  The communicator id travels over the existing process group."""
    import ctypes as C

    from . import capi
    hl = capi.hip_lib()

    def agreed(ok, what):
        """True when every rank succeeded; otherwise every rank raises the same way (nobody is left waiting in a collective)."""
        if world_size > 1:
            flags = [None] * world_size
            dist.all_gather_object(flags, (bool(ok), what))
        else:
            flags = [(bool(ok), what)]
        bad = [(k, w) for k, (good, w) in enumerate(flags) if not good]
        if bad:
            raise capi.RtError(capi.RTHIP_E_COMM, "rank %d: %s" % bad[0])

    # every rank proves it can load RCCL before anyone enters ncclCommInitRank (a rank that cannot would leave the rest hanging)
    buf = C.create_string_buffer(capi.RTHIP_COMM_ID_BYTES)
    rc = hl.rthip_comm_unique_id(buf)
    agreed(rc == 0, "rthip_comm_unique_id: " + capi.error_string(rc))
    box = [buf.raw if rank == 0 else None]
    if world_size > 1:
        dist.broadcast_object_list(box, src=0)
    rc = hl.rthip_comm_init(renderer.context, box[0], int(rank), int(world_size))
    agreed(rc == 0, "rthip_comm_init: " + capi.error_string(rc))


def agree_traversal_mode(renderer, dist, rank, world_size):
    """Every rank measures its traversal mode on a sample of ITS tiles (rthip_begin -> calibrateRefill), so ranks could settle
    on different modes: the frame's bits would not change, the balance would.  Rank 0's measured choice is broadcast and
    fixed on every rank (rthip_set_refill_mode).  Call after the first begin_frame.  Returns the agreed mode."""
    mode = [int(renderer.refill_mode()[0]) if rank == 0 else None]
    if world_size > 1:
        dist.broadcast_object_list(mode, src=0)
    renderer.set_option("refillMode", mode[0])
    return mode[0]


# ---------------------------------------------------------------------------------------------------------------------
# Dynamic tile ownership: over-decomposed tiles handed out from one shared cursor, so a GPU whose
# tiles were cheap takes more of them instead of idling through the others' tail.  A whole tile (all its samples) is
# still rendered by one rank, so per-pixel accumulation order -- and therefore every bit of the frame -- is unchanged.

class TileQueue:
    """A cursor over the row-major tile list shared by all ranks through a torch.distributed store (the store's `add` is
    atomic).  No data-path collective is involved: ranks only ever touch the counter."""

    KEY = "rt_tile_cursor"

    def __init__(self, store, num_tiles, batch):
        assert batch >= 1
        self.store, self.num_tiles, self.batch = store, int(num_tiles), int(batch)
        self.key = self.KEY

    def new_frame(self, frame_index):
        """A fresh cursor for the next frame (every rank calls it with the same index; no reset race: a new key)."""
        self.key = "%s_%d" % (self.KEY, int(frame_index))

    def next_batch(self):
        """The next `batch` tile ids as a uint32 array, fewer at the end of the list, None when the list is exhausted."""
        end = int(self.store.add(self.key, self.batch))
        start = end - self.batch
        if start >= self.num_tiles:
            return None
        return np.arange(start, min(end, self.num_tiles), dtype=np.uint32)


def make_queue_store(dist, rank, world_size, host="127.0.0.1"):
    """A TCPStore of its own for the tile cursor (rank 0 serves it); the port is agreed through the process group."""
    import socket
    import torch.distributed as tdist
    port = [0]
    if rank == 0:
        with socket.socket() as s:
            s.bind((host, 0))
            port[0] = s.getsockname()[1]
    if world_size > 1:
        dist.broadcast_object_list(port, src=0)
    return tdist.TCPStore(host, port[0], world_size, is_master=(rank == 0), wait_for_workers=False)


def gather_dynamic(local, tile_ids, rank, world_size, dist, tile_pixels):
    """Gathers per-rank tile-major buffers of DIFFERENT lengths to rank 0: the tile-id lists first (padded with -1), then the
    pixel buffers padded to the longest.  Returns [(tile_ids, buffer)] per rank on rank 0, None elsewhere."""
    import torch
    tile_ids = np.asarray(tile_ids, dtype=np.int64)
    counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world_size)]
    mine = torch.tensor([len(tile_ids)], dtype=torch.int64, device=local.device)
    if world_size > 1:
        dist.all_gather(counts, mine)
    else:
        counts = [mine]
    most = max(1, max(int(c.item()) for c in counts))
    ids = torch.full((most,), -1, dtype=torch.int64, device=local.device)
    ids[: len(tile_ids)] = torch.from_numpy(tile_ids).to(local.device)
    buf = torch.zeros((most * tile_pixels, 3), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    if world_size == 1:
        return [(tile_ids.astype(np.uint32), buf[: len(tile_ids) * tile_pixels])]
    id_bufs = [torch.empty_like(ids) for _ in range(world_size)] if rank == 0 else None
    px_bufs = [torch.empty_like(buf) for _ in range(world_size)] if rank == 0 else None
    dist.gather(ids, id_bufs, dst=0)
    dist.gather(buf, px_bufs, dst=0)
    if rank != 0:
        return None
    out = []
    for c, i, b in zip(counts, id_bufs, px_bufs):
        k = int(c.item())
        out.append((i[:k].cpu().numpy().astype(np.uint32), b[: k * tile_pixels]))
    return out
