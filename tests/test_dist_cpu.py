"""N > 1 data-parallel path on CPU: world_size-2 gloo processes shard a batch, embed their slice and
all-gather pooled embeddings; the result must equal the single-process result row for row."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from avex_amd import dist as adist


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 256, 2048, 2051):
        for w in (1, 2, 3, 8):
            spans = [adist.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert adist.shard_bounds(2048, 3, 8) == (768, 1024)        # BASELINE config C4: 256 clips per GPU


def _embed(wav: torch.Tensor) -> torch.Tensor:
    """Stand-in for model.extract_embeddings(...): any per-clip function works, clips are independent."""
    feats = torch.stack([wav.mean(1), wav.std(1), wav.abs().max(1)[0], (wav ** 2).mean(1)], dim=1)
    return feats.float()


def _worker(rank, world, port, n_clips, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = adist.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    wav = torch.randn(n_clips, 4000, generator=g)
    full = adist.extract_embeddings_sharded(_embed, wav)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [8, 7, 1])
def test_two_rank_gloo_all_gather(tmp_path, n_clips):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, n_clips, str(tmp_path)), nprocs=2, join=True)
    g = torch.Generator().manual_seed(0)
    ref = _embed(torch.randn(n_clips, 4000, generator=g)).numpy()
    for r in range(2):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.shape == ref.shape and np.array_equal(got, ref)       # every rank holds every row, in clip order


def test_single_process_is_a_no_op():
    wav = torch.randn(3, 100)
    assert torch.equal(adist.extract_embeddings_sharded(_embed, wav), _embed(wav))
