"""Camera-sharded reference-sheet rendering across the GPUs of one node.

The reference renders the cameras of a sheet in a sequential loop on one device
(/root/reference/signerf/datasetgenerator/datasetgenerator.py:517-519, 331); the loop body has no cross-iteration state
before the diffuser call (:558), so cameras shard embarrassingly: camera i -> rank i % world_size, one process per GPU,
weights replicated.  The only exchange is ONE all-gather of the finished [H,W,4] (rgb + median depth) tiles
(10.24 MB per 800x800 camera) -- RCCL over xGMI with the "nccl" backend, gloo on CPU for the tests.
"""

from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def shard_indices(n_items: int, world_size: int, rank: int) -> List[int]:
    """Round-robin ownership: item i belongs to rank i % world_size."""
    return list(range(rank, n_items, world_size))


def _stage_through_host(t: Tensor, group=None) -> bool:
    """gloo has no all-gather for device tensors: with that backend (single-GPU debugging, the world-2 tests that put both ranks on
    one GPU) GPU tiles are staged through the host.  RCCL ("nccl") gathers device memory directly over xGMI."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class TileGather:
    """An all-gather (or gather-to-root) of tiles in flight (``gather_tiles_async``).  ``wait()`` orders the caller's stream behind the
    collective (RCCL: a stream wait, the host does not block) and returns the tiles in item order -- ``None`` on the ranks that are
    not the destination of a gather-to-root."""

    def __init__(self, work, gathered: Optional[Tensor], n_items: int, device=None):
        self._work, self._gathered, self._n_items, self._device = work, gathered, n_items, device

    def wait(self) -> Optional[Tensor]:
        if self._work is not None:
            self._work.wait()
            self._work = None
            if self._gathered is not None and self._gathered.is_cuda:
                # the buffer may have been allocated on a frame's stream (FrameStreams) and is read from the waiting stream from here on
                self._gathered.record_stream(torch.cuda.current_stream(self._gathered.device))
        if self._gathered is None:
            return None
        if self._device is not None:  # staged through the host (gloo): back to the GPU the tiles came from
            self._gathered, self._device = self._gathered.to(self._device), None
        g = self._gathered
        if g.dim() == 4:  # single process: already [n_items, H, W, C]
            return g
        world, per = g.shape[:2]
        # gathered[r, k] is item r + k * world  ->  item-major order is the transpose
        return g.transpose(0, 1).reshape(world * per, *g.shape[2:])[: self._n_items].contiguous()


GATHER_STRATEGIES = ("all_gather", "p2p", "all_to_all")


def default_strategy() -> str:
    """SIGNERF_GATHER in the environment selects the exchange of the sheet / generator loops ("all_gather" when unset)."""
    import os

    s = os.environ.get("SIGNERF_GATHER", "all_gather")
    if s not in GATHER_STRATEGIES:
        raise ValueError(f"SIGNERF_GATHER={s!r}: one of {GATHER_STRATEGIES} expected")
    return s


class _Works:
    """Several point-to-point work handles (and the buffers they read) waited for as one."""

    def __init__(self, works, *keep):
        self._works, self._keep = list(works), keep

    def wait(self):
        for w in self._works:
            w.wait()
        self._works, self._keep = [], ()


def gather_tiles_async(local_tiles: Tensor, n_items: int, group=None, dst: Optional[int] = None, strategy: str = "all_gather") -> TileGather:
    """Starts the all-gather of per-rank tiles and returns at once, so that the next camera's render overlaps the exchange
    (the collective runs on RCCL's own stream).  Keep ``local_tiles`` unmodified until ``wait()``.

    ``dst``: gather to that rank only -- in the generator loop only rank 0 composes the sheets and talks to the
    diffuser (/root/reference/signerf/datasetgenerator/datasetgenerator.py:558), so the other ranks need no tiles: 1/world of the
    bytes of the all-gather cross the links, and every sender uses its own xGMI link to the root.

    ``strategy`` -- how the bytes travel (xGMI is point to point, 7 links x ~153 GB/s per GPU, so a RING
    all-gather of 8 tiles is bound by one link for 7 hops while 7 direct pushes use 7 links at once):
      "all_gather"   ``all_gather_into_tensor`` / ``gather``: RCCL picks the algorithm (ring or tree over the xGMI mesh); one collective call
      "p2p"          every rank SENDS its tile straight to each peer and RECEIVES each peer's tile (``batch_isend_irecv``: world - 1 sends
                     and receives per rank, one per link); with ``dst`` only the sends to / receives at the root
      "all_to_all"   ``all_to_all_single`` of the tile replicated per destination: the same direct pushes as one RCCL call
                     (costs a world-fold staging copy of the tile)
    The result is identical; `bench.py --gpus N` reports the exposed time of each so that the first multi-GPU run compares them."""
    if strategy not in GATHER_STRATEGIES:
        raise ValueError(f"gather strategy {strategy!r}: one of {GATHER_STRATEGIES} expected")
    if not (dist.is_available() and dist.is_initialized()):
        assert local_tiles.shape[0] == n_items
        return TileGather(None, local_tiles, n_items)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = (n_items + world - 1) // world
    pad = per - local_tiles.shape[0]
    if pad > 0:
        local_tiles = torch.cat([local_tiles, local_tiles.new_zeros((pad, *local_tiles.shape[1:]))], dim=0)
    local_tiles = local_tiles.contiguous()
    device = None
    if _stage_through_host(local_tiles, group):
        device, local_tiles = local_tiles.device, local_tiles.cpu()
    glob = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    tile_shape = local_tiles.shape[1:]
    if strategy == "p2p":
        if dst is None:
            gathered = local_tiles.new_empty((world, per, *tile_shape))
            gathered[rank].copy_(local_tiles)
            ops = [dist.P2POp(dist.isend, local_tiles, glob(r), group) for r in range(world) if r != rank]
            ops += [dist.P2POp(dist.irecv, gathered[r], glob(r), group) for r in range(world) if r != rank]
            works = dist.batch_isend_irecv(ops) if ops else []
            return TileGather(_Works(works, local_tiles), gathered, n_items, device)
        if rank == dst:
            gathered = local_tiles.new_empty((world, per, *tile_shape))
            gathered[rank].copy_(local_tiles)
            ops = [dist.P2POp(dist.irecv, gathered[r], glob(r), group) for r in range(world) if r != rank]
            works = dist.batch_isend_irecv(ops) if ops else []
            return TileGather(_Works(works), gathered, n_items, device)
        works = dist.batch_isend_irecv([dist.P2POp(dist.isend, local_tiles, glob(dst), group)])
        return TileGather(_Works(works, local_tiles), None, n_items, None)
    if strategy == "all_to_all" and dst is None:
        gathered = local_tiles.new_empty((world, per, *tile_shape))
        send = local_tiles.unsqueeze(0).expand(world, per, *tile_shape).contiguous()   # the tile once per destination
        work = dist.all_to_all_single(gathered.view(world * per, *tile_shape), send.view(world * per, *tile_shape), group=group, async_op=True)
        return TileGather(_KeepAlive(work, send), gathered, n_items, device)
    # "all_gather" (and "all_to_all" towards one root, which is a plain gather)
    if dst is None:
        gathered = local_tiles.new_empty((world, per, *tile_shape))
        work = dist.all_gather_into_tensor(gathered.view(world * per, *tile_shape), local_tiles, group=group, async_op=True)
        return TileGather(work, gathered, n_items, device)
    if rank == dst:
        gathered = local_tiles.new_empty((world, per, *tile_shape))
        work = dist.gather(local_tiles, list(gathered.unbind(0)), dst=glob(dst), group=group, async_op=True)
        return TileGather(work, gathered, n_items, device)
    work = dist.gather(local_tiles, None, dst=glob(dst), group=group, async_op=True)
    return TileGather(_KeepAlive(work, local_tiles), None, n_items, None)


class _KeepAlive:
    """A collective's work handle together with the send buffer it reads (a host staging copy must outlive the send)."""

    def __init__(self, work, *tensors):
        self._work, self._tensors = work, tensors

    def wait(self):
        self._work.wait()
        self._tensors = ()


def gather_tiles(local_tiles: Tensor, n_items: int, group=None, dst: Optional[int] = None, strategy: str = "all_gather") -> Optional[Tensor]:
    """All-gather per-rank tiles back into item order.

    local_tiles: [n_local, H, W, C] -- this rank's tiles for items rank, rank+world, ... (n_local may differ by one
    between ranks).  Returns [n_items, H, W, C] on every rank (``dst`` given: on that rank only, None elsewhere).
    """
    return gather_tiles_async(local_tiles, n_items, group, dst, strategy).wait()


class FrameStreams:
    """Consecutive, independent frames on alternating HIP streams.

    A render launch ends with a partly filled last round of waves (800x800: 10 000 waves on the chip's 3072 wave slots = 3.26 rounds),
    and the next launch of the same stream cannot start before it has drained.  Frames of different cameras do not depend on each
    other, so issuing them on alternating streams lets the head of frame i + 1 fill the slots the tail of frame i leaves idle:
    measured r02 (tools/two_stream_probe.py) 2.89 -> 2.68 ms per 800x800x64 frame, 14.68 -> 14.40 ms per 1080p nerfacto frame; a third
    stream adds nothing.  Renders of one handle are unordered and re-entrant (include/signerf_hip.h), results are bit-identical.

        with FrameStreams(device) as fs:          # a CPU device (the gloo tests) degrades to plain in-order execution
            for k, cam in enumerate(cams):
                with fs.frame(k):
                    tiles.append(fs.keep(render(cam)))
        # leaving the block joins: the caller's stream now waits for every frame (also when a render raised)
    """

    _pool: dict = {}   # (device index, k) -> side stream k of that device: one set per process, not one per call (every new stream
    #                    opens its own pool of the caching allocator)

    def __init__(self, device, frames_in_flight: int = 2):
        self._cur = None
        self._streams: List = []
        if device is not None and torch.device(device).type == "cuda" and frames_in_flight > 1 and torch.cuda.is_available():
            device = torch.device(device)
            index = device.index if device.index is not None else torch.cuda.current_device()
            self._cur = torch.cuda.current_stream(device)
            for k in range(frames_in_flight):
                if (index, k) not in FrameStreams._pool:
                    FrameStreams._pool[(index, k)] = torch.cuda.Stream(device=device)
                self._streams.append(FrameStreams._pool[(index, k)])
            for st in self._streams:
                st.wait_stream(self._cur)  # inputs prepared on the caller's stream (uploads, camera tensors)

    def __enter__(self) -> "FrameStreams":
        return self

    def __exit__(self, exc_type, exc, tb) -> bool:
        self.join()   # also on an exception inside a frame: nothing the caller already holds may be read ahead of the side streams
        return False

    def frame(self, k: int):
        import contextlib

        return torch.cuda.stream(self._streams[k % len(self._streams)]) if self._streams else contextlib.nullcontext()

    def keep(self, t: Optional[Tensor]) -> Optional[Tensor]:
        """Marks a tensor produced inside ``frame`` as consumed on the caller's stream (caching-allocator bookkeeping)."""
        if t is not None and self._streams and t.is_cuda:
            t.record_stream(self._cur)
        return t

    def join(self) -> None:
        for st in self._streams:
            self._cur.wait_stream(st)


def render_cameras_sharded(render_fn: Callable[[int], Tuple[Tensor, Tensor]], n_cameras: int, group=None, device=None,
                           frames_in_flight: int = 2, dst: Optional[int] = None, strategy: Optional[str] = None) -> Optional[Tensor]:
    """Renders cameras round-robin over the ranks and all-gathers the tiles.

    render_fn(i) -> (rgb [H,W,3], depth [H,W,1]) for camera i, on this rank's device.  ``device``: this rank's GPU -- its cameras are
    then issued on ``frames_in_flight`` alternating streams (FrameStreams); None keeps one stream.
    Returns [n_cameras, H, W, 4] (rgb ++ depth) on every rank, identical to a single-rank run (``dst`` given: gather to that rank
    only, None on the others).
    """
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    mine = shard_indices(n_cameras, world, rank)
    tiles = []
    with FrameStreams(device, frames_in_flight) as fs:
        for k, i in enumerate(mine):
            with fs.frame(k):
                rgb, depth = render_fn(i)
                tiles.append(fs.keep(torch.cat([rgb, depth], dim=-1)))
    if tiles:
        local = torch.stack(tiles, dim=0)
    else:  # fewer cameras than ranks: learn the tile shape from rank 0's broadcast
        local = None
    if world > 1:
        shape = torch.tensor(list(local.shape[1:]) if local is not None else [0, 0, 0], dtype=torch.int64)
        dev = local.device if local is not None else _default_device()
        shape = shape.to(dev if dist.get_backend(group) != "gloo" else "cpu")
        dist.broadcast(shape, src=0, group=group)
        if local is None:
            local = torch.zeros((0, *shape.tolist()), dtype=torch.float32, device=dev)
    return gather_tiles(local, n_cameras, group, dst, strategy or default_strategy())


def _default_device():
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def render_views(model, cameras, generator_config, group=None, frames_in_flight: int = 2, dst: Optional[int] = None,
                 render_camera_fn=None) -> Optional[Tensor]:
    """BASELINE.json configs[4]: the per-view work of the dataset-generator loops
    (/root/reference/signerf/datasetgenerator/datasetgenerator.py:331-338 and :517-519): for every camera, render ->
    mask -> condition (``render_camera``, aabb mode), sharded round-robin over the ranks, tiles all-gathered (``dst``: gathered to
    that rank only).
    -> [n_cameras, H, W, 5] = rgb (3) ++ mask (1, 0/1) ++ condition (1) on every rank.  The diffusion call that follows
    each view in the reference is a remote HTTP service and stays where it is (rank 0)."""
    from . import datasetgenerator

    render_camera = render_camera_fn or datasetgenerator.render_camera

    def render_fn(i: int):
        rgb, mask, cond = render_camera(generator_config, model, cameras[i])
        return rgb, torch.cat([mask.to(rgb.dtype), cond], dim=-1)

    return render_cameras_sharded(render_fn, len(cameras), group, device=getattr(model, "device", None), frames_in_flight=frames_in_flight,
                                  dst=dst)


def render_reference_sheet(model, cameras, group=None) -> Tensor:
    """Row (e): every camera of ``cameras`` (a batched ``signerf_amd.Cameras`` on this rank's GPU) rendered by its owner
    rank through the reference's two calls, tiles all-gathered.  -> [n_cameras, H, W, 4]."""

    def render_fn(i: int):
        cam = cameras[i]
        bundle = cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb)
        out = model.get_outputs_for_camera_ray_bundle(bundle)
        return out["rgb"], out["depth"]

    return render_cameras_sharded(render_fn, len(cameras), group, device=getattr(model, "device", None))


# ----------------------------------------------------------------------------------------------------------------------
# Fallback of SURVEY §8(e): fewer cameras than GPUs, or one big frame -- contiguous row blocks of ONE image per rank.
# ----------------------------------------------------------------------------------------------------------------------
def row_blocks(height: int, world_size: int, align: int = 8) -> List[Tuple[int, int]]:
    """[row0, row1) per rank: contiguous blocks whose sizes are multiples of `align` rows (the kernels' 8x8 pixel tiles) except
    the last non-empty one, as equal as that allows; trailing ranks may get an empty block."""
    units = (height + align - 1) // align
    per, extra = divmod(units, world_size)
    out, r = [], 0
    for k in range(world_size):
        n = (per + (1 if k < extra else 0)) * align
        r1 = min(height, r + n)
        out.append((r, r1))
        r = r1
    return out


def gather_rows(local_rows: Tensor, blocks: Sequence[Tuple[int, int]], group=None) -> Tensor:
    """All-gather row blocks of one image back into [H, W, C] on every rank (blocks padded to the tallest for the collective)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_rows
    world = dist.get_world_size(group)
    tallest = max(r1 - r0 for r0, r1 in blocks)
    W, Cn = local_rows.shape[1], local_rows.shape[2]
    pad = tallest - local_rows.shape[0]
    if pad > 0:
        local_rows = torch.cat([local_rows, local_rows.new_zeros((pad, W, Cn))], dim=0)
    local_rows = local_rows.contiguous()
    device = None
    if _stage_through_host(local_rows, group):
        device, local_rows = local_rows.device, local_rows.cpu()
    gathered = local_rows.new_empty((world * tallest, W, Cn))
    dist.all_gather_into_tensor(gathered, local_rows, group=group)
    if device is not None:
        gathered = gathered.to(device)
    return torch.cat([gathered[k * tallest : k * tallest + (r1 - r0)] for k, (r0, r1) in enumerate(blocks)], dim=0)


def render_camera_row_sharded(model, camera, group=None) -> Tensor:
    """One camera split into contiguous row blocks over the ranks (weights replicated), tiles all-gathered:
    -> [H, W, 4] (rgb ++ median depth) on every rank, identical to a single-rank render -- rays are independent (§8(e));
    the only per-chunk quantity of the path, the clip range of `expected_depth`, is not part of this tile."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    bundle = camera.generate_rays(camera_indices=0, aabb_box=model.render_aabb)  # ray generation is ~20 us for a full frame
    H = bundle.origins.shape[0]
    blocks = row_blocks(H, world)
    r0, r1 = blocks[rank]
    if r1 > r0:
        out = model.get_outputs_for_camera_ray_bundle(bundle._map(lambda t: t[r0:r1].contiguous()))
        local = torch.cat([out["rgb"], out["depth"]], dim=-1)
    else:
        local = bundle.origins.new_zeros((0, bundle.origins.shape[1], 4))
    return gather_rows(local, blocks, group)
