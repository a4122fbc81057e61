"""Dataset split / standardisation conventions (experiments/utils.py:21-79) on a synthetic .mat file."""
import numpy as np
import pytest
import torch
from scipy.io import savemat

from simplex_gp_amd import data


def test_prepare_dataset_matches_reference_conventions(tmp_path):
    rng = np.random.default_rng(0)
    raw = rng.standard_normal((1000, 6)) * np.array([1, 5, 0.1, 3, 2, 10]) + np.array([0, 1, -2, 3, 0, 5])
    savemat(str(tmp_path / "toy.mat"), {"data": raw})
    splits = {m: (x, y) for m, x, y in data.prepare_dataset("toy", uci_data_dir=str(tmp_path))}
    assert list(splits) == ["train", "val", "test"]
    assert [splits[m][0].shape[0] for m in splits] == [640, 160, 200]          # 0.8*0.8, 0.8*0.2, 0.2
    assert splits["train"][0].shape[1] == 5 and splits["train"][1].dim() == 1   # last column is the target
    xt = torch.as_tensor(raw[:640, :-1], dtype=torch.float32)
    want = (xt - xt.mean(0, keepdim=True)) / (xt.std(0, keepdim=True) + 2e-6)
    assert torch.allclose(splits["train"][0], want, atol=1e-5)
    assert abs(float(splits["train"][1].mean())) < 1e-5 and abs(float(splits["train"][1].std()) - 1) < 1e-4
    # validation / test use the TRAINING statistics (so they are not exactly standardised)
    xv = torch.as_tensor(raw[640:800, :-1], dtype=torch.float32)
    assert torch.allclose(splits["val"][0], (xv - xt.mean(0, keepdim=True)) / (xt.std(0, keepdim=True) + 2e-6), atol=1e-5)
    with pytest.raises(FileNotFoundError):
        list(data.prepare_dataset("elevators", uci_data_dir=str(tmp_path)))
    with pytest.raises(AssertionError):
        list(data.prepare_dataset(None, uci_data_dir=str(tmp_path)))
