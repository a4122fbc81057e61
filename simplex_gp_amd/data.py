"""Dataset preparation with the reference's conventions.

experiments/utils.py:21-44, :47-79: a UCI regression set is one `.mat` file whose
"data" matrix holds the features with the target in the last column; rows are
taken in file order, the first 64 % train, the next 16 % validation, the last
20 % test (0.8 x 0.8 / 0.8 x 0.2 / 0.2); features and target are standardised
with the TRAINING mean and std (std + 1e-6, divided by std + 1e-6 again).
The UCI files are not redistributable and not in this repo; `prepare_arrays`
applies the same split / standardisation to any (x, y).
"""
import os

import torch


def split_bounds(n, train_val_split=0.8):
    n_train_val = int(train_val_split * n)
    n_train = int(train_val_split * n_train_val)
    return n_train, n_train_val


def prepare_arrays(x, y, device=None, train_val_split=0.8):
    """{mode: (x, y)} for mode in train / val / test, standardised with the training statistics."""
    x = torch.as_tensor(x, dtype=torch.float32, device=device)
    y = torch.as_tensor(y, dtype=torch.float32, device=device)
    n_train, n_train_val = split_bounds(x.shape[0], train_val_split)
    parts = {"train": slice(0, n_train), "val": slice(n_train, n_train_val), "test": slice(n_train_val, None)}
    x_mean, x_std = x[parts["train"]].mean(0, keepdim=True), x[parts["train"]].std(0, keepdim=True) + 1e-6
    y_mean, y_std = y[parts["train"]].mean(0, keepdim=True), y[parts["train"]].std(0, keepdim=True) + 1e-6
    return {m: (((x[s] - x_mean) / (x_std + 1e-6)).contiguous(), ((y[s] - y_mean) / (y_std + 1e-6)).contiguous())
            for m, s in parts.items()}


def prepare_dataset(dataset, uci_data_dir=None, device=None, train_val_split=0.8):
    """Load `<uci_data_dir>/<dataset>.mat` (or $DATADIR/uci/...) and split / standardise it.
    Yields (mode, x, y) in the order train, val, test, like the reference generator."""
    from scipy.io import loadmat
    if uci_data_dir is None and os.environ.get("DATADIR") is not None:
        uci_data_dir = os.path.join(os.environ["DATADIR"], "uci")
    assert dataset is not None and uci_data_dir is not None, f'Select a dataset from "{uci_data_dir}"'
    path = dataset if str(dataset).endswith(".mat") else os.path.join(str(uci_data_dir), f"{dataset}.mat")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: the UCI .mat files are not part of this repository")
    data = torch.as_tensor(loadmat(path)["data"], dtype=torch.float32)
    splits = prepare_arrays(data[:, :-1], data[:, -1], device=device, train_val_split=train_val_split)
    for mode in ("train", "val", "test"):
        yield (mode,) + splits[mode]
