"""Input pipeline: collate parity against the reference's own collate function (tests/golden/collate.npz), shard
round trip, sampler / loader semantics (CPU), and the uint8 -> normalised fp32 image kernel + a loader-fed train step (GPU)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vacnic_amd import synthetic          # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
CASES = {"mixed": (11, 6, None), "no_faces_anywhere": (12, 3, "nofaces"), "single_name_row": (13, 4, "noname")}   # as oracle/make_golden.py
KEYS = ("article_ids", "caption_ids", "names_art_ids", "names_ids", "names_ids_flatten", "face_emb")


def case_samples(case):
    seed, n, edit = CASES[case]
    samples = synthetic.make_samples(n, seed=seed)
    for sm in samples:
        if edit == "nofaces":
            sm["face_emb"] = sm["face_emb"][:0]
        if edit == "noname":
            sm["names_ids"] = np.array([[0, 50266, 2]], dtype=np.int64)
    return samples


@pytest.mark.parametrize("case", list(CASES))
def test_collate_bit_exact_against_reference_collate(case):
    from oracle import vacnic_oracle as O
    from vacnic_amd import data
    g = np.load(os.path.join(G, "collate.npz"))
    samples = case_samples(case)
    got, rest = data.collate(samples), O.collate_restated(samples)
    for k in KEYS:
        ref = g[f"{case}:{k}"]
        assert got[k].shape == ref.shape and got[k].dtype == ref.dtype and np.array_equal(got[k], ref), (case, k)
        assert np.array_equal(rest[k].numpy(), ref), (case, k, "oracle restatement")
    assert np.array_equal(got["image_u8"].astype(np.float32), g[f"{case}:img_tensor"])


def test_shard_round_trip_and_sampler(tmp_path):
    from vacnic_amd import data
    samples = synthetic.make_samples(23, seed=5)
    path = os.path.join(str(tmp_path), "s.vshard")
    with data.ShardWriter(path) as w:
        for s in samples:
            w.add(s)
    rd = data.ShardReader(path)
    assert len(rd) == 23
    for i in (0, 7, 22):
        for k, v in samples[i].items():
            assert np.array_equal(rd[i][k], v) and rd[i][k].shape == v.shape, (i, k)
    a, b = data.collate([rd[i] for i in range(5)]), data.collate(samples[:5])
    assert all(np.array_equal(a[k], b[k]) for k in a)
    # DistributedSampler semantics: pad to a multiple of world, stride by world, per-epoch permutation shared by all ranks
    ld = [data.PrefetchLoader(rd, 4, device="cpu", rank=r, world=3, seed=9, depth=2) for r in range(3)]
    idx = [l.indices() for l in ld]
    assert all(len(i) == 8 for i in idx)
    assert sorted(np.concatenate(idx).tolist())[:23] != [] and set(np.concatenate(idx).tolist()) == set(range(23))
    for l in ld:
        l.set_epoch(1)
    assert not np.array_equal(ld[0].indices(), idx[0])
    batches = list(ld[0])
    assert len(batches) == len(ld[0]) == 2
    again = list(ld[0])
    for (x, _), (y, _) in zip(batches, again):
        assert all(torch.equal(x[k], y[k]) for k in x), "same (seed, epoch, rank) -> same batches"
    first = data.collate([rd[int(i)] for i in ld[0].indices()[:4]])
    assert torch.equal(batches[0][0]["article_ids"], torch.from_numpy(first["article_ids"]))


@pytest.mark.gpu
def test_image_u8_normalize_bit_exact():
    from vacnic_amd import kernels as K
    from vacnic_amd.data import CLIP_MEAN, CLIP_STD
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (5, 3, 37, 29), generator=g, dtype=torch.uint8)
    flip = torch.tensor([0, 1, 0, 1, 1], dtype=torch.uint8)
    t = img.float().div(255)                                                     # ToTensor
    t = torch.where(flip.bool()[:, None, None, None], t.flip(-1), t)             # RandomHorizontalFlip (given coins)
    want = (t - torch.tensor(CLIP_MEAN)[None, :, None, None]) / torch.tensor(CLIP_STD)[None, :, None, None]   # Normalize
    got = K.image_u8_normalize(img.cuda(), flip.cuda(), CLIP_MEAN, CLIP_STD)
    assert torch.equal(got.cpu(), want)
    assert torch.equal(K.image_u8_normalize(img.cuda()).cpu(), (img.float().div(255) - torch.tensor(CLIP_MEAN)[None, :, None, None]) / torch.tensor(CLIP_STD)[None, :, None, None])


@pytest.mark.gpu
def test_loader_feeds_train_step(tmp_path):
    from vacnic_amd import data
    from vacnic_amd.config import ClipVisionConfig, VacnicConfig
    from vacnic_amd.training import FusedAdamW, TrainArgs, build_models, train_step
    cfg = VacnicConfig(d_model=768, encoder_layers=1, decoder_layers=1, encoder_attention_heads=12, decoder_attention_heads=12,
                       encoder_ffn_dim=3072, decoder_ffn_dim=3072, enc_fusion_layer=[0], dim_common=768, clip_width=768, dropout=0.1)
    vcfg = ClipVisionConfig(width=768, layers=1, patch_size=16, image_size=32, output_dim=64)
    path = os.path.join(str(tmp_path), "s.vshard")
    with data.ShardWriter(path) as w:
        for s in synthetic.make_samples(12, seed=3):
            w.add(s)
    model, guide, _ = build_models(cfg, vcfg, init="synthetic", seed=0)
    args = TrainArgs(num_training_steps=10, lr_bart=1e-4)
    opt = FusedAdamW(model.arena, lr=1e-4, num_warmup_steps=1, num_training_steps=10)
    loader = data.PrefetchLoader(data.ShardReader(path), 4, seed=1)
    n = 0
    for batch, ready in loader:
        assert batch["img_tensor"].shape == (4, 3, 32, 32) and batch["img_tensor"].dtype == torch.float32
        assert batch["face_emb"].shape[1] >= 1 and batch["names_ids"].shape[2] >= 3
        out4 = train_step(model, guide, opt, batch, args, ready)
        assert np.isfinite(out4.tolist()).all()
        n += 1
    assert n == 3
