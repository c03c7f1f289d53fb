"""Drop-in `dataset.base` (reference: dataset/base.py:4-19): the interface the dataset classes of the path implement."""
from torch.utils.data import Dataset


class DiscreteData(Dataset):
    """A dataset of token sequences.  Subclasses say how many items they hold (`get_length`), how wide an item is
    (`get_dim`) and hand out items by index.  (The reference marks the three as abstract without an ABC metaclass, so
    nothing is enforced at construction there either; here a forgotten override fails at the call.)"""

    def get_length(self):
        raise NotImplementedError(f'{type(self).__name__}.get_length')

    def get_dim(self):
        raise NotImplementedError(f'{type(self).__name__}.get_dim')

    def __getitem__(self, i):
        raise NotImplementedError(f'{type(self).__name__}.__getitem__')
