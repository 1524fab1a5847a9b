from .dataset import ArrayDataset, Dataset  # noqa: F401
from .dataloader import (BatchSampler, DataLoader, RandomSampler, Sampler, SequentialSampler,  # noqa: F401
                         data_loader)
