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
