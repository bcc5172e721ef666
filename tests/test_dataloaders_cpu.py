"""Host logic of the data path: sampler order / coverage, transforms on numpy and on tensors drawing the same
random stream.  (The h5 reader needs h5py and the datasets: not testable in this image.)"""
import numpy as np
import torch

from dycon_paper_replication_amd.dataloaders import brats19 as D


def test_two_stream_sampler_labelled_first_and_epoch_length():
    labelled, unlabelled = list(range(25)), list(range(25, 250))            # BraTS19 labelnum = 25 (train_DyCON_BraTS19.py:250-256)
    s = D.TwoStreamBatchSampler(labelled, unlabelled, batch_size=8, secondary_batch_size=4)
    assert len(s) == 25 // 4
    np.random.seed(1337)
    batches = list(s)
    assert len(batches) == len(s)
    seen = []
    for b in batches:
        assert len(b) == 8 and all(i < 25 for i in b[:4]) and all(i >= 25 for i in b[4:])
        seen += list(b[:4])
    assert len(set(seen)) == len(seen) == 24                                 # one pass over the primary indices, no repeats
    np.random.seed(1337)
    assert [tuple(b) for b in D.TwoStreamBatchSampler(labelled, unlabelled, 8, 4)] == [tuple(b) for b in batches]   # seeded stream


def test_transforms_numpy_and_tensor_agree():
    rng = np.random.default_rng(0)
    img = rng.standard_normal((40, 36, 30)).astype(np.float32)
    lab = (rng.random((40, 36, 30)) > 0.8).astype(np.uint8)
    for tf in (D.RandomCrop((32, 32, 16)), D.RandomRotFlip(), D.CenterCrop((32, 32, 16)), D.RandomCrop((48, 48, 32))):   # last: pad path
        np.random.seed(7)
        a = tf({"image": img, "label": lab})
        np.random.seed(7)
        b = tf({"image": torch.from_numpy(img), "label": torch.from_numpy(lab)})
        np.testing.assert_array_equal(a["image"], b["image"].numpy())
        np.testing.assert_array_equal(a["label"], b["label"].numpy())
    np.random.seed(3)
    out = D.RandomCrop((48, 48, 32))({"image": img, "label": lab})
    assert out["image"].shape == (48, 48, 32) and out["label"].shape == (48, 48, 32)
    t = D.ToTensor()({"image": img, "label": lab})
    assert t["image"].shape == (1, 40, 36, 30) and t["image"].dtype == torch.float32 and t["label"].dtype == torch.int64
    np.random.seed(5)
    n1 = D.RandomNoise()({"image": img, "label": lab})["image"]
    assert np.abs(n1 - img).max() <= 0.2 + 1e-6


def test_sagittal_to_axial():
    rng = np.random.default_rng(1)
    img = rng.standard_normal((6, 5, 4)).astype(np.float32)
    lab = (rng.random((6, 5, 4)) > 0.5).astype(np.uint8)
    a = D.SagittalToAxial()({"image": img, "label": lab})
    assert a["image"].shape == (4, 5, 6) and a["image"][1, 2, 3] == img[3, 2, 1] and a["label"][0, 4, 5] == lab[5, 4, 0]
    b = D.SagittalToAxial()({"image": torch.from_numpy(img), "label": torch.from_numpy(lab)})
    np.testing.assert_array_equal(a["image"], b["image"].numpy())
    np.testing.assert_array_equal(a["label"], b["label"].numpy())
    import pytest
    with pytest.raises(ValueError):
        D.SagittalToAxial()({"image": img, "label": lab[:5]})


def test_pancreas_isles_transforms_and_samplers(tmp_path):
    """the two other loader modules of the reference (code/dataloaders/pancreas.py, isles22.py): shared transforms re-exported,
    CreateOnehotLabel / ToTensor with 'onehot_label', RandomRot, ThreeStreamBatchSampler, list-file handling of the datasets"""
    from dycon_paper_replication_amd.dataloaders import isles22 as I
    from dycon_paper_replication_amd.dataloaders import pancreas as P
    assert P.RandomCrop is D.RandomCrop and I.RandomRotFlip is D.RandomRotFlip and I.TwoStreamBatchSampler is D.TwoStreamBatchSampler
    rng = np.random.default_rng(2)
    img = rng.standard_normal((10, 9, 8))
    lab = (rng.random((10, 9, 8)) > 0.7).astype(np.uint8)
    s = P.CreateOnehotLabel(2)({"image": img, "label": lab})
    assert s["onehot_label"].shape == (2, 10, 9, 8) and s["onehot_label"].dtype == np.float32
    np.testing.assert_array_equal(s["onehot_label"][1], lab.astype(np.float32))
    np.testing.assert_array_equal(s["onehot_label"].sum(0), np.ones_like(lab, dtype=np.float32))
    t = P.ToTensor()(s)
    assert t["image"].shape == (1, 10, 9, 8) and t["image"].dtype == torch.float32
    assert t["label"].dtype == torch.int64 and t["onehot_label"].dtype == torch.int64
    st = I.CreateOnehotLabel(2)({"image": torch.from_numpy(img), "label": torch.from_numpy(lab)})
    np.testing.assert_array_equal(st["onehot_label"].numpy(), s["onehot_label"])
    # RandomRot: angle from np.random.randint(-20, 20), nearest neighbour, shape kept; angle 0 is the identity
    from scipy import ndimage
    np.random.seed(11)
    angle = np.random.randint(-20, 20)
    np.random.seed(11)
    r = I.RandomRot()({"image": img, "label": lab})
    np.testing.assert_array_equal(r["label"], ndimage.rotate(lab, angle, order=0, reshape=False))
    assert r["image"].shape == img.shape and set(np.unique(r["label"])) <= {0, 1}
    # ThreeStreamBatchSampler as the reference behaves: its two primary groupers iterate the SAME permutation array independently, so a
    # batch is [primary group | secondary group | the same primary group] and an epoch has len(primary) // primary_batch_size batches
    import itertools
    smp = I.ThreeStreamBatchSampler(list(range(10)), list(range(10, 40)), batch_size=4, secondary_batch_size=2)
    np.random.seed(3)
    batches = list(smp)
    assert len(batches) == len(smp) == 5
    assert all(len(b) == 6 and b[:2] == b[4:] and all(i < 10 for i in b[:2]) and all(i >= 10 for i in b[2:4]) for b in batches)
    assert sorted(i for b in batches for i in b[:2]) == list(range(10))
    # the same draws, restated with the reference's own construction (grouper = zip of n references to one iterator)
    np.random.seed(3)
    prim = np.random.permutation(list(range(10)))
    sec = itertools.chain.from_iterable(np.random.permutation(list(range(10, 40))) for _ in itertools.count())
    grouper = lambda it, n: zip(*[iter(it)] * n)  # noqa: E731
    expect = [a + b + c for a, b, c in zip(grouper(prim, 2), grouper(sec, 2), grouper(prim, 2))]
    assert [tuple(int(i) for i in b) for b in batches] == [tuple(int(i) for i in e) for e in expect]
    # dataset list handling (no h5py needed until a sample is read)
    (tmp_path / "train.list").write_text("case_a.h5\ncase_b.h5\ncase_c.h5\n")
    (tmp_path / "test.list").write_text("case_z.h5\n")
    ds = P.Pancreas(str(tmp_path), split="train", num=2)
    assert len(ds) == 2 and ds.image_list == ["case_a.h5", "case_b.h5"]
    assert len(P.Pancreas(str(tmp_path), split="val")) == 1
    (tmp_path / "val.list").write_text("s1\ns2\n\n")
    (tmp_path / "s1.h5").write_bytes(b"")
    assert I.ISLESDataset(str(tmp_path), split="val").sample_list == ["s1.h5"]      # listed cases without a file are skipped
    import pytest
    with pytest.raises(FileNotFoundError):
        I.ISLESDataset(str(tmp_path), split="nope")
