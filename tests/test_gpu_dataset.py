"""Scenes read from disk drive the training step on the GPU: the loader's batches are what the step expects."""
import os

import pytest
import torch

import hivt_gan_amd as hv

pytestmark = pytest.mark.gpu


def test_training_steps_from_scene_files(tmp_path):
    root = str(tmp_path)
    d = os.path.join(root, "train_processed")
    os.makedirs(d)
    for i in range(6):
        hv.save_scene(hv.make_scene(100 + i), os.path.join(d, f"{i:03d}.pt"))
    data = hv.SceneDataset(root, "train")
    torch.manual_seed(3)
    model = hv.HiVT(embed_dim=64, num_heads=8, dropout=0.1, num_temporal_layers=1, num_global_layers=1, use_gan=True,
                    rotate=True).to("cuda:0")
    model.train()
    trainer = hv.GanTrainer(model)
    before = torch.cat([p.detach().flatten() for p in model.parameters()]).clone()
    seen = 0
    for batch in hv.batches(data, 3, shuffle=True, seed=1):
        assert batch.num_graphs == 3
        logs = trainer.step(batch.to("cuda:0"))
        assert "skipped" not in logs or float(logs["skipped"]) == 0.0
        assert all(bool(torch.isfinite(v).all()) for v in logs.values())
        seen += 1
    assert seen == 2
    after = torch.cat([p.detach().flatten() for p in model.parameters()])
    assert bool(torch.isfinite(after).all()) and float((after - before).abs().max()) > 0
    # the same scenes through the in-memory generator give the same batch tensors
    direct = hv.collate([hv.make_scene(100 + i) for i in range(3)])
    from_disk = hv.collate([data[i] for i in range(3)])
    for k in direct.keys():
        if torch.is_tensor(direct[k]):
            assert torch.equal(direct[k], from_disk[k]), k
