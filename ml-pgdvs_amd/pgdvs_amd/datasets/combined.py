"""Mirror of ``pgdvs.datasets.combined.CombinedDataset`` (pgdvs/datasets/combined.py:31-80): one index space
over the datasets named in ``dataset_list[mode]``, in sorted-name order so that every worker sees the same
order.  Loaders mirrored here are used directly; the two that are out of scope (``nvidia_vis``,
``dycheck_iphone_eval``) resolve to the reference's own classes when that package is importable."""
import bisect
import importlib

from torch.utils.data import Dataset

_MIRRORED = {
    "nvidia_eval": ("pgdvs_amd.datasets.nvidia_eval", "NvidiaDynEvaluationDataset"),
    "nvidia_eval_pure_geo": ("pgdvs_amd.datasets.nvidia_eval", "NvidiaDynPureGeoEvaluationDataset"),
    "mono_vis": ("pgdvs_amd.datasets.mono_vis", "MonoVisualizationDataset"),
}
_UPSTREAM = {
    "nvidia_vis": ("pgdvs.datasets.nvidia_vis", "NvidiaDynVisualizationDataset"),
    "dycheck_iphone_eval": ("pgdvs.datasets.dycheck_iphone_eval", "DyCheckiPhoneEvaluationDataset"),
}


def dataset_class(name: str):
    if name in _MIRRORED:
        mod, cls = _MIRRORED[name]
    elif name in _UPSTREAM:
        mod, cls = _UPSTREAM[name]
    else:
        raise KeyError(f"unknown dataset {name!r}; known: {sorted(_MIRRORED) + sorted(_UPSTREAM)}")
    try:
        return getattr(importlib.import_module(mod), cls)
    except ImportError as e:
        raise ImportError(f"dataset {name!r} is not mirrored in pgdvs_amd and the reference package is not importable ({e})") from e


class CombinedDataset(Dataset):
    def __init__(self, *, data_root, dataset_list, mode="train", max_hw=-1, rgb_range="0_1", use_aug=False, dataset_specifics={}):
        assert mode in ["train", "eval", "vis"], mode
        if mode in ["eval", "vis"]:
            use_aug = False
        self.datasets = {
            name: dataset_class(name)(data_root=data_root, max_hw=max_hw, rgb_range=rgb_range, use_aug=use_aug, mode=mode,
                                      **dataset_specifics[name])
            for name in dataset_list[mode]}
        self._names = sorted(self.datasets)
        self._ends = []  # cumulative lengths in sorted-name order
        total = 0
        for name in self._names:
            total += len(self.datasets[name])
            self._ends.append(total)

    def __len__(self):
        return self._ends[-1] if self._ends else 0

    def __getitem__(self, index):
        if not 0 <= index < len(self):
            raise IndexError(index)
        k = bisect.bisect_right(self._ends, index)
        return self.datasets[self._names[k]][index - (self._ends[k - 1] if k else 0)]
