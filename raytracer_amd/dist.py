"""Multi-GPU partitioning of the trace loop: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" on CPU for tests).

Every pixel, sub-sample and frame is independent given the read-only scene (this is what
the reference's parallel_for over pixels exploits, Raytracer/Raytracer.cpp:672), so the
scene is replicated and the only exchange step is gathering finished framebuffer pieces on
rank 0.  Two partitions:

  bands   one frame split into interleaved bands of rows: band b belongs to rank b % world.
          Interleaving evens out the very different cost of sky, floor and glass pixels.
  frames  a frame sequence split round-robin: frame f belongs to rank f % world.

The random streams are keyed by the global pixel index, so an image does not depend on how
it was partitioned.
"""
import torch
import torch.distributed as dist


def band_edges(height, world, band_rows=64, bands_per_rank=None):
    """Row boundaries of the bands of a frame.  With `bands_per_rank` the frame is cut into
    world * bands_per_rank bands of (nearly) equal height, so that every rank owns the same
    number of bands and, within a row, of rows; otherwise bands are `band_rows` rows high."""
    if bands_per_rank:
        n = min(height, world * bands_per_rank)
        return [(b * height) // n for b in range(n + 1)]
    edges = list(range(0, height, band_rows)) + [height]
    return edges


def band_rects(width, height, world, rank, band_rows=64, bands_per_rank=None):
    """(x0, y0, w, h) of the bands owned by `rank`, top to bottom in frame order: band b
    belongs to rank b % world."""
    edges = band_edges(height, world, band_rows, bands_per_rank)
    return [(0, edges[b], width, edges[b + 1] - edges[b]) for b in range(len(edges) - 1) if b % world == rank]


def frames_for_rank(n_frames, world, rank):
    return list(range(rank, n_frames, world))


def _gather_to_rank0(local, world, rank, group=None):
    """Gather equally shaped tensors on rank 0 (a list of `world` tensors there, None elsewhere)."""
    if world == 1:
        return [local]
    out = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, out, dst=0, group=group)
    return out


def render_frame_bands(render_rect, width, height, device, band_rows=64, group=None, comm_device=None,
                       bands_per_rank=None, after_bands=None):
    """Render one frame cooperatively.

    render_rect(x0, y0, w, h, out) must accumulate the rectangle's float32 RGB into `out`
    (a [h, w, 3] tensor on `device`, zero-initialised here).  Returns the assembled
    [height, width, 3] frame on rank 0 and None on the other ranks.  `comm_device` is where
    the gather runs when the process group cannot move `device` tensors (gloo with GPU
    framebuffers); the frame is then returned there.  Bands: see band_edges().
    render_rect may only queue its work (asynchronous scene handles on several streams);
    `after_bands()` is then called once all of this rank's bands are queued and must make the
    current stream wait for them.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    edges = band_edges(height, world, band_rows, bands_per_rank)
    n_bands = len(edges) - 1
    per_rank = (n_bands + world - 1) // world
    tallest = max(edges[b + 1] - edges[b] for b in range(n_bands))
    # every rank packs its bands into the same padded shape so that one gather moves them
    packed = torch.zeros((per_rank, tallest, width, 3), dtype=torch.float32, device=device)
    mine = band_rects(width, height, world, rank, band_rows, bands_per_rank)
    for k, (x0, y0, w, h) in enumerate(mine):
        render_rect(x0, y0, w, h, packed[k, :h])
    if after_bands is not None:
        after_bands()
    if comm_device is not None and torch.device(comm_device) != packed.device:
        packed = packed.to(comm_device)
    parts = _gather_to_rank0(packed, world, rank, group)
    if rank != 0:
        return None
    frame = torch.empty((height, width, 3), dtype=torch.float32, device=packed.device)
    for r in range(world):
        for k, (x0, y0, w, h) in enumerate(band_rects(width, height, world, r, band_rows, bands_per_rank)):
            frame[y0:y0 + h] = parts[r][k, :h]
    return frame


def rank_rows(height, world, rank, group=8):
    """Frame rows owned by `rank` when the frame is dealt out in groups of `group` rows, group g
    to rank g % world: (first row, number of rows, stride between the rank's groups).  The rows
    are first + (r // group) * stride + r % group for r in range(number): one tile with
    tile_row_group = group, tile_row_stride = stride (rtx.h), rendered by one call."""
    n_groups = (height + group - 1) // group
    mine = len(range(rank, n_groups, world))
    rows = mine * group
    if mine and (rank + (mine - 1) * world) == n_groups - 1:      # this rank owns the last group, which may be short
        rows -= n_groups * group - height
    return rank * group, rows, world * group


_buffers = {}


def _buffer(tag, shape, dtype, device):
    """A tensor kept between frames (no allocation, no fill per frame); contents are whatever
    the last frame left."""
    key = (tag, tuple(shape), dtype, str(device))
    if key not in _buffers:
        _buffers[key] = torch.zeros(shape, dtype=dtype, device=device)
    return _buffers[key]


def render_frame_rows(render_rows, width, height, device, group=8, pg=None, comm_device=None, after=None, finish=None,
                      zero=True):
    """Render one frame cooperatively, every rank its interleaved groups of rows in ONE call.

    render_rows(y0, rows, group, stride, out) renders the tile (0, y0, width, rows) with
    tile_row_group = group and tile_row_stride = stride into `out` ([rows, width, 3] float32 on
    `device`, packed rows).  With `zero` the buffer is cleared first (a render call adds to it);
    a caller that renders with RTX_FLAG_OVERWRITE passes zero=False and saves the fill.  Finer than
    bands and without their many short launches: a rank's share is one tile of 1/world of the frame.
    `after()` is called once the rank's work is queued and must make the current stream wait for it.
    `finish(rows_tensor)` may turn the rank's float rows into what is gathered instead -- e.g. the
    presented 8-bit rows, a quarter of the bytes (a driver that accumulates passes keeps the float
    rows on their rank and only ever needs the presented ones on rank 0).
    Rank 0 returns the assembled [height, width, 3] frame (row 0 = bottom row, as the renderer
    numbers them), the others None.  All buffers are kept between frames; rank 0 places the
    gathered rows with ONE strided copy: rank r's k-th group is group r + k * world of the frame.

    THE RETURNED TENSOR IS A VIEW OF ONE OF THOSE KEPT BUFFERS: it is valid until the next call of
    this function in the process, which overwrites it.  A caller that keeps frames (a list of
    frames, a writer that works behind the renderer) must `.clone()` what it keeps."""
    world = dist.get_world_size(pg) if dist.is_initialized() else 1
    rank = dist.get_rank(pg) if dist.is_initialized() else 0
    n_groups = (height + group - 1) // group
    per_rank = (n_groups + world - 1) // world          # groups of the busiest rank
    most = per_rank * group
    packed = _buffer("packed", (most, width, 3), torch.float32, device)
    y0, rows, stride = rank_rows(height, world, rank, group)
    if rows:
        if zero:
            packed[:rows].zero_()
        render_rows(y0, rows, group, stride, packed[:rows])
    if after is not None:
        after()
    local = finish(packed) if finish is not None else packed
    if comm_device is not None and torch.device(comm_device) != local.device:
        local = local.to(comm_device)
    if world == 1 and most == height:
        return local                                     # one rank owns the rows in frame order
    gathered = _buffer("gathered", (world, most, width, 3), local.dtype, local.device) if rank == 0 else None
    if world == 1:
        gathered[0].copy_(local)
    else:
        dist.gather(local, list(gathered.unbind(0)) if rank == 0 else None, dst=0, group=pg)
    if rank != 0:
        return None
    frame = _buffer("frame", (per_rank * world * group, width, 3), local.dtype, local.device)
    # frame row (k * world + r) * group + j  <-  rank r, packed row k * group + j
    frame.view(per_rank, world, group, width, 3).copy_(gathered.view(world, per_rank, group, width, 3).transpose(0, 1))
    return frame[:height]


def render_sequence_frames(render_frame, n_frames, width, height, device, group=None):
    """Render frames 0..n_frames-1 round-robin.  render_frame(f, out) fills the [H, W, 3]
    float32 tensor `out`.  Rank 0 returns the list of all frames in order, others None."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    rounds = (n_frames + world - 1) // world
    frames = [None] * n_frames if rank == 0 else None
    buf = torch.zeros((height, width, 3), dtype=torch.float32, device=device)
    for k in range(rounds):
        f = k * world + rank
        buf.zero_()
        if f < n_frames:
            render_frame(f, buf)
        parts = _gather_to_rank0(buf, world, rank, group)
        if rank == 0:
            for r in range(world):
                if k * world + r < n_frames:
                    frames[k * world + r] = parts[r].clone()
    return frames
