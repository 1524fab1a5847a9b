"""Map-style dataset base class (reference: DeepFlows/utils/data/dataset.py:1-9): indexable, sized."""


class Dataset:
    def __getitem__(self, index):
        raise NotImplementedError

    def __len__(self):
        raise NotImplementedError


class ArrayDataset(Dataset):
    """(data[index], target[index]) over two equally long arrays -- what `data_loader()` wraps its arguments in
    (the reference's local `TrainSet`, utils/data/dataloader.py:129-137).  `index` may be an int or a list of ints
    (a batch of indices from the BatchSampler: NumPy fancy indexing, one copy)."""

    def __init__(self, data, target):
        if len(data) != len(target):
            raise ValueError(f"data and target differ in length: {len(data)} vs {len(target)}")
        self.data, self.target = data, target

    def __getitem__(self, index):
        return self.data[index], self.target[index]

    def __len__(self):
        return len(self.data)
