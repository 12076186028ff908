import torch

from distributed_machine_learning_amd.config import CIFAR_MEAN, CIFAR_STD
from distributed_machine_learning_amd.data import (BatchLoader, ShardedSampler, synthetic_cifar)
from distributed_machine_learning_amd.data.transforms import _augment_reference
from distributed_machine_learning_amd.parallel.flat import plan_buckets


def test_rank_strided_shards_reassemble_global_batch():
    ds = synthetic_cifar(1000, 1)
    world, bs = 4, 64
    firsts = []
    for r in range(world):
        s = ShardedSampler(len(ds), r, world)
        firsts.append(s.indices()[:bs])
        assert s.indices()[0].item() == r and s.indices()[1].item() == r + world
    union = torch.cat(firsts).sort().values
    assert torch.equal(union, torch.arange(world * bs))


def test_sampler_pads_by_wrapping():
    s = ShardedSampler(10, 2, 4)
    assert len(s) == 3
    assert s.indices().tolist() == [2, 6, 0]


def test_loader_batches_and_last_partial():
    ds = synthetic_cifar(10, 3)
    batches = list(BatchLoader(ds, 4))
    assert [b[0].shape[0] for b in batches] == [4, 4, 2]
    assert len(BatchLoader(ds, 4)) == 3


def test_augment_matches_per_image_pipeline():
    ds = synthetic_cifar(6, 5)
    aug = torch.tensor([[4, 4, 0], [0, 0, 0], [8, 8, 1], [3, 6, 1], [4, 4, 1], [7, 1, 0]],
                       dtype=torch.int32)
    out = _augment_reference(ds.images, aug)
    mean = torch.tensor(CIFAR_MEAN).view(3, 1, 1)
    std = torch.tensor(CIFAR_STD).view(3, 1, 1)
    for i in range(6):
        img = ds.images[i].permute(2, 0, 1).float() / 255.0
        padded = torch.zeros(3, 40, 40)
        padded[:, 4:36, 4:36] = img
        oy, ox, fl = aug[i].tolist()
        crop = padded[:, oy:oy + 32, ox:ox + 32]
        if fl:
            crop = crop.flip(-1)
        ref = (crop - mean) / std
        assert torch.allclose(out[i], ref, atol=1e-6)
    plain = _augment_reference(ds.images, None)
    assert torch.allclose(plain[0], (ds.images[0].permute(2, 0, 1).float() / 255 - mean) / std,
                          atol=1e-6)


def test_bucket_plan_vgg11_25mb():
    sizes = [1728, 64, 73728, 128, 294912, 256, 589824, 256, 1179648, 512, 2359296, 512,
             2359296, 512, 2359296, 512, 5120, 10]
    buckets = plan_buckets([4 * s for s in sizes], 25 << 20)
    # buckets are contiguous, cover everything, and come back-to-front
    assert buckets[0][1] == len(sizes) and buckets[-1][0] == 0
    for (lo, hi), (lo2, hi2) in zip(buckets, buckets[1:]):
        assert hi2 == lo and lo2 < hi2
    # first bucket closes right after it passes 1 MiB: fc1 + conv8 bias + conv8 weight
    assert buckets[0] == (14, 18)
    assert len(buckets) == 3
