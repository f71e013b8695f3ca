"""Multi-GPU driver pieces: one process per GPU, clips sharded by contiguous blocks, ONE exchange at the end.

The reference has no distributed code.  Clips are independent (no cross-sample op anywhere,
BatchNorm is in eval mode), so ranks never talk during generation; `gather_sequences` is the single RCCL all-gather
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests) of the generated pose sequences.
Determinism contract: the result for global clip k does not depend on the world size — greedy decode is a pure
function of the clip, and the stochastic sampler's Philox subsequence is the global clip index (clip_index0).
"""
import torch
import torch.distributed as dist


def shard_range(n_clips, rank, world):
    """Contiguous block of ceil(n/world) clips for `rank` (last ranks may get fewer / none): (start, stop)."""
    per = (n_clips + world - 1) // world
    start = min(rank * per, n_clips)
    return start, min(start + per, n_clips)


def gather_sequences(local, n_total=None):
    """all-gather of (n_local, T, C) sequences -> (n_total, T, C) on every rank, in global clip order.

    Ranks may hold different n_local (ragged tail): shards are padded to the largest block for the collective and
    trimmed afterwards.  Without an initialised process group this is the identity.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    out = torch.empty((world * nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    parts = [out[r * nmax:r * nmax + counts[r]] for r in range(world)]
    res = torch.cat(parts, 0)
    if n_total is not None:
        assert res.shape[0] == n_total, (res.shape, n_total)
    return res


def generate_sharded(generate_fn, mfcc, ids, batch=32):
    """Run `generate_fn(mfcc_block, ids_block, clip_index0)` over THIS rank's shard in batches; no collective here.

    mfcc (N,T,64) / ids (N,) are the GLOBAL inputs (every rank holds them, or at least its own block).
    generate_fn returns (codes, poses (n,T',C)) for a block.  Returns (local poses (n_local,T',C), (start, stop)); hand
    `local` to `gather_sequences` for the one exchange.  A rank whose shard is empty (fewer clips than ranks, or a
    ragged tail) returns a (0,T',C) tensor with the right trailing shape, dtype and device for the collective:
    generate_fn is asked for them with a zero-clip block (`mfcc[0:0]`), which the HIP wrapper answers without a launch
    (`output_shape`), a stand-in by returning an empty result.
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = mfcc.shape[0]
    a, b = shard_range(n, rank, world)
    outs = []
    for s in range(a, b, batch):
        e = min(s + batch, b)
        _, poses = generate_fn(mfcc[s:e], ids[s:e], s)
        outs.append(poses)
    if outs:
        local = torch.cat(outs, 0)
    else:
        local = _empty_like_output(generate_fn, mfcc, ids)
    return local, (a, b)


def _empty_like_output(generate_fn, mfcc, ids):
    """(0,T',C) tensor on the device / in the dtype generate_fn produces, for a rank with no clips."""
    shape_fn = getattr(generate_fn, "output_shape", None)
    if shape_fn is not None:
        tail, dtype, device = shape_fn(mfcc)
        return torch.zeros((0,) + tuple(tail), dtype=dtype, device=device)
    _, probe = generate_fn(mfcc[0:0], ids[0:0], 0)
    if probe.ndim < 2 or probe.shape[0] != 0:
        raise RuntimeError("generate_fn must answer a zero-clip block with a (0,T',C) tensor (or expose output_shape)")
    return probe


_side_streams = {}


def _side_stream(device, k=0):
    """library-created side stream k of the device (each its own hardware queue and scratch arena)"""
    from . import _lib
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if (idx, k) not in _side_streams:
        _side_streams[(idx, k)] = _lib.create_streams(1, idx)[0]
    return _side_streams[(idx, k)]


def whole_body_local(body, face, mfcc, ids, wav, face_ids, mode=None, seed=0, clip_index0=0, batch_body=32, batch_face=64,
                     stand=False, overlap=True):
    """Whole-body generation of THIS rank's clips (no collective): body path in batches of `batch_body`, face path in
    batches of `batch_face`, (n, Tf, 265) rows assembled on the GPU (`pose_index.assemble_full` = demo.py:207-229 +
    part2full).  mfcc (n,T,64), ids (n,), wav (n,S) 16 kHz samples, face_ids (n,4); `clip_index0` = global index of
    the first clip (Philox subsequences).  n = 0 gives a (0, Tf, 265) tensor on the current device."""
    from . import _lib
    from .pose_index import assemble_full
    n = mfcc.shape[0]
    frames = wav.shape[1] * 30 // 16000                       # smplx_face.py:203
    mode = _lib.TS_SAMPLE_PHILOX if mode is None else mode
    if n == 0:
        return torch.zeros((0, frames, 265), dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
    poses, faces = [], []
    # the two generators are independent until the assembly: the body path (its autoregressive chain is latency-bound and leaves most
    # of the matrix pipe idle) goes to a side stream and runs under the face generator's GEMMs; both join before the assembly
    cur = torch.cuda.current_stream()
    side = _side_stream(cur.device) if overlap else cur
    if overlap:
        side.wait_stream(cur)
    with torch.cuda.stream(side):
        for s in range(0, n, batch_body):
            e = min(s + batch_body, n)
            poses.append(body.generate_batch(mfcc[s:e], ids[s:e], mode=mode, seed=seed, clip_index0=clip_index0 + s)[1])
    # face batches alternate between the current stream and a second side stream: a GEMM's partly filled last round and the gap
    # between one batch's dependent launches are filled by the other batch's workgroups (tools/face_streams.py: 59.4 -> 56.6 ms per
    # batch of 64 with two in flight); the generator keeps its scratch per stream, a batch's rows do not depend on the stream it ran on
    face_streams = [cur, _side_stream(cur.device, 1)] if overlap and n > batch_face else [cur]
    for st in face_streams[1:]:
        st.wait_stream(cur)
    for i, s in enumerate(range(0, n, batch_face)):
        e = min(s + batch_face, n)
        with torch.cuda.stream(face_streams[i % len(face_streams)]):
            faces.append(face.generator.run(wav[s:e], face_ids[s:e], frames))
    for st in face_streams[1:]:
        cur.wait_stream(st)
    for i, t in enumerate(faces):
        if face_streams[i % len(face_streams)] is not cur:
            t.record_stream(cur)
    if overlap:
        cur.wait_stream(side)
        for t in poses:
            t.record_stream(cur)
    return assemble_full(torch.cat(poses, 0), torch.cat(faces, 0), stand=stand)


def whole_body_sharded(body, face, mfcc, ids, wav, face_ids, mode=None, seed=0, batch_body=32, batch_face=64, stand=False):
    """BASELINE configs[4]: whole-body generation of N clips sharded over the ranks, one all-gather at the end.

    body / face: the `nets.s2g_body_pixel` / `nets.s2g_face` wrappers of this rank (full weight replicas).
    mfcc (N,T,64), ids (N,) int64 speaker indices, wav (N,S) 16 kHz samples, face_ids (N,4) one-hot / zero float vectors:
    the GLOBAL inputs (a rank only touches its own block).  Returns (all_rows (N,Tf,265) on every rank, (start, stop) of
    this rank's block); a rank with an empty block contributes a (0,Tf,265) shard.
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = mfcc.shape[0]
    a, b = shard_range(n, rank, world)
    local = whole_body_local(body, face, mfcc[a:b], ids[a:b], wav[a:b], face_ids[a:b], mode=mode, seed=seed,
                             clip_index0=a, batch_body=batch_body, batch_face=batch_face, stand=stand)
    return gather_sequences(local, n), (a, b)
