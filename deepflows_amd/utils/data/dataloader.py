"""In-memory batch loader with the reference's interface (DeepFlows/utils/data/dataloader.py:6-139): samplers that
yield indices, a BatchSampler that groups them -- keeping the ragged last batch unless `drop_last` --, a DataLoader
whose iterator indexes the dataset with one index LIST per batch, and the `data_loader(X, y, ...)` helper every
example script calls (test/ResNet_CIFAR10_cuda.py:162-163, test/CNN_CIFAR10_cuda.py:58-59, test/MLP_MNIST_cuda.py:39-40).

Behaviour that callers (and seeded runs) depend on, kept on purpose:
  * `RandomSampler` draws ONE `numpy.random.permutation(len(dataset))` from the GLOBAL generator each time it is
    iterated (:30-31) -- the draw happens when the first batch is requested, or at `iter(loader)` when
    `prefetch_size > 0` (the iterator fills its look-ahead queue in its constructor, :60-66);
  * a batch is `dataset[list_of_indices]`, made contiguous with `np.ascontiguousarray` when `as_contiguous`;
  * `prefetch_size` batches are indexed ahead of the consumer (a plain in-process queue, no threads);
  * the loader itself has no length in the reference; scripts use `len(loader.batch_sampler)`.  `len(loader)` is
    provided here as well and returns the same number.
The batch order under a fixed seed is pinned by index sequences the reference's own classes produced
(oracle/gen_epoch_golden.py -> tests/golden/g12_epoch.npz).
"""
from collections import deque

import numpy as np

from .dataset import ArrayDataset, Dataset  # noqa: F401


class Sampler:
    """Iterable over dataset indices (or, for BatchSampler, over lists of them)."""

    def __init__(self, source=None):
        self.source = source

    def __iter__(self):
        raise NotImplementedError

    def __len__(self):
        return len(self.source)


class SequentialSampler(Sampler):
    def __init__(self, dataset):
        super().__init__(dataset)
        self.dataset = dataset

    def __iter__(self):
        return iter(range(len(self.dataset)))


class RandomSampler(Sampler):
    def __init__(self, dataset):
        super().__init__(dataset)
        self.dataset = dataset

    def __iter__(self):
        def walk():
            # global generator, one draw per pass, made when the first index is asked for
            for index in np.random.permutation(len(self.dataset)).tolist():
                yield index
        return walk()


class BatchSampler(Sampler):
    def __init__(self, sampler, batch_size, drop_last):
        super().__init__(sampler)
        if int(batch_size) <= 0:
            raise ValueError("batch_size must be positive")
        self.sampler, self.batch_size, self.drop_last = sampler, int(batch_size), bool(drop_last)

    def __iter__(self):
        def batches():
            group = []
            for index in self.sampler:              # a RandomSampler draws its permutation HERE, on the first request
                group.append(index)
                if len(group) == self.batch_size:
                    yield group
                    group = []
            if group and not self.drop_last:
                yield group
        return batches()

    def __len__(self):
        full, rest = divmod(len(self.sampler), self.batch_size)
        return full if (self.drop_last or rest == 0) else full + 1


class _BatchIterator:
    def __init__(self, loader):
        self._loader = loader
        self._indices = iter(loader.batch_sampler)
        self._ahead = deque()
        self._top_up()

    def _load(self, index_list):
        batch = self._loader.dataset[index_list]
        if self._loader.as_contiguous:
            x, y = batch
            batch = (np.ascontiguousarray(x), np.ascontiguousarray(y))
        return batch

    def _top_up(self):
        while len(self._ahead) < self._loader.prefetch_size:
            index_list = next(self._indices, None)
            if index_list is None:
                return
            self._ahead.append(self._load(index_list))

    def __iter__(self):
        return self

    def __next__(self):
        if self._ahead:
            batch = self._ahead.popleft()
            self._top_up()
            return batch
        return self._load(next(self._indices))       # StopIteration ends the pass


class DataLoader:
    def __init__(self, dataset, batch_size=1, shuffle=False, drop_last=False, prefetch_size: int = 0,
                 as_contiguous: bool = True):
        self.dataset = dataset
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.drop_last = drop_last
        self.prefetch_size = max(0, int(prefetch_size))
        self.as_contiguous = as_contiguous
        self.sampler = RandomSampler(dataset) if shuffle else SequentialSampler(dataset)
        self.batch_sampler = BatchSampler(self.sampler, batch_size, drop_last)

    def __iter__(self):
        return _BatchIterator(self)

    def __len__(self):
        return len(self.batch_sampler)


def data_loader(X, y, batch_size, shuffle=False, prefetch_size: int = 0, as_contiguous: bool = True):
    """DataLoader over two arrays; the ragged last batch is kept (drop_last=False), as in the reference (:128-139)."""
    return DataLoader(ArrayDataset(X, y), batch_size, shuffle, prefetch_size=prefetch_size, as_contiguous=as_contiguous)
