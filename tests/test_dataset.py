"""On-disk scenes: flat-dict files, the reference's shape sanitation (``datasets/nuscenes_dataset.py:55-103``),
sorted / truncated listing (``:20-41``) and the rank-sharded batch iterator."""
import os

import pytest
import torch

import hivt_gan_amd as hv
from hivt_gan_amd import dataset as ds


def write_split(root, split, count, first=0):
    d = os.path.join(root, f"{split}_processed")
    os.makedirs(d)
    for i in range(count):
        ds.save_scene(hv.make_scene(first + i), os.path.join(d, f"scene_{i:04d}.pt"))
    return d


def same_scene(a, b):
    assert set(a.keys()) == set(b.keys())
    for k in a.keys():
        if torch.is_tensor(a[k]):
            assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


def test_round_trip_and_listing(tmp_path):
    root = str(tmp_path)
    d = write_split(root, "train", 5)
    open(os.path.join(d, "notes.txt"), "w").close()  # only .pt files are scenes
    full = ds.SceneDataset(root, "train")
    assert len(full) == 5 and full.processed_file_names == sorted(full.processed_file_names)
    for i in range(5):
        same_scene(full[i], hv.make_scene(i))
    assert len(ds.SceneDataset(root, "train", max_samples=3)) == 3
    with pytest.raises(FileNotFoundError):
        ds.SceneDataset(root, "val")
    seen = []
    tagged = ds.SceneDataset(root, "train", transform=lambda s: seen.append(int(s.num_nodes)) or s)
    tagged[2]
    assert seen == [int(hv.make_scene(2).num_nodes)]
    batch = full.collate_fn([full[0], full[1]])
    same_scene(batch, hv.make_batch(2))


def test_sanitize_shapes_and_errors():
    s = hv.make_scene(3)
    s["lane_actor_index"] = torch.tensor([4, 1])
    s["edge_index"] = torch.tensor([0, 1])
    out = ds.sanitize(s)
    assert out["lane_actor_index"].shape == (2, 1) and out["edge_index"].shape == (2, 1)
    s = hv.make_scene(3)
    s["lane_actor_index"] = torch.empty(0, dtype=torch.long)
    s["lane_actor_vectors"] = torch.empty(0)
    s["lane_vectors"] = torch.empty(0)
    s["edge_index"] = torch.empty(0, dtype=torch.long)
    out = ds.sanitize(s)
    assert out["lane_actor_index"].shape == (2, 0) and out["edge_index"].shape == (2, 0)
    assert out["lane_actor_vectors"].shape == (0, 2) and out["lane_vectors"].shape == (0, 2)
    s = hv.make_scene(3)
    s["lane_actor_index"] = None
    s["lane_actor_vectors"] = None
    s["lane_vectors"] = None
    out = ds.sanitize(s)
    assert out["lane_actor_index"].shape == (2, 0) and out["lane_actor_index"].dtype == torch.long
    assert out["lane_actor_vectors"].shape == (0, 2) and out["lane_vectors"].shape == (0, 2)
    for key, bad, text in (("lane_actor_index", torch.zeros(3, 4, dtype=torch.long), "Bad lane_actor_index shape"),
                           ("lane_actor_vectors", torch.zeros(4, 3), "Bad lane_actor_vectors shape"),
                           ("lane_vectors", torch.zeros(6), "Bad lane_vectors shape"),
                           ("edge_index", torch.zeros(3, 3, dtype=torch.long), "Bad edge_index shape")):
        s = hv.make_scene(3)
        s[key] = bad
        with pytest.raises(ValueError, match=text):
            ds.sanitize(s)


class Thing:  # stands for a data object of some library class, pickled whole
    pass


def test_foreign_pickles_are_refused(tmp_path):
    path = os.path.join(str(tmp_path), "obj.pt")
    torch.save(Thing(), path)
    with pytest.raises(RuntimeError, match="not a flat dict"):
        ds.load_scene(path)
    torch.save([1, 2, 3], path)
    with pytest.raises(RuntimeError, match="expected a dict"):
        ds.load_scene(path)


def test_sharded_batches_cover_every_scene(tmp_path):
    root = str(tmp_path)
    write_split(root, "train", 7)
    data = ds.SceneDataset(root, "train")
    sizes = [int(hv.make_scene(i).num_nodes) for i in range(7)]
    plain = list(ds.batches(data, 3))
    assert [b.num_graphs for b in plain] == [3, 3, 1]
    assert [int(b.num_nodes) for b in plain] == [sum(sizes[0:3]), sum(sizes[3:6]), sizes[6]]
    assert [b.num_graphs for b in ds.batches(data, 3, drop_last=True)] == [3, 3]
    a = [int(b.num_nodes) for b in ds.batches(data, 2, shuffle=True, seed=5, epoch=1)]
    b = [int(b.num_nodes) for b in ds.batches(data, 2, shuffle=True, seed=5, epoch=1)]
    c = [int(b.num_nodes) for b in ds.batches(data, 2, shuffle=True, seed=5, epoch=2)]
    assert a == b and sum(a) == sum(sizes) and a != c
    # two ranks: equal batch counts (the order wraps around), every scene seen by some rank
    per_rank = [[int(b.num_nodes) for b in ds.batches(data, 1, shuffle=True, seed=1, rank=r, world_size=2)] for r in (0, 1)]
    assert len(per_rank[0]) == len(per_rank[1]) == 4
    assert set(per_rank[0]) | set(per_rank[1]) == set(sizes)


def test_out_of_range_indices_are_refused():
    """An index that addresses no agent / lane must raise on the host (the device kernels index unchecked)."""
    import pytest
    from hivt_gan_amd.dataset import sanitize
    scene = hv.make_scene(2)
    bad = scene.clone()
    bad["edge_index"][0, 0] = int(scene.num_nodes)
    with pytest.raises(ValueError, match="edge_index entries"):
        sanitize(bad)
    bad = scene.clone()
    bad["lane_actor_index"][0, 0] = int(scene["lane_vectors"].shape[0])
    with pytest.raises(ValueError, match="lane ids"):
        sanitize(bad)
    bad = scene.clone()
    bad["lane_actor_index"][1, 0] = -1
    with pytest.raises(ValueError, match="actor ids"):
        sanitize(bad)
    sanitize(scene.clone())
