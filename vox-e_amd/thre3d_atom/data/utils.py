"""Data-loading helpers kept from the reference's thre3d_atom/data/utils.py (`infinite_dataloader` :44-47).
The torchvision image transform of that module is not needed: `PosedImagesDataset` decodes with PIL / numpy."""
from typing import Any, Iterable, Iterator


def infinite_dataloader(data_loader: Iterable[Any]) -> Iterator[Any]:
    """Cycle over a (re-iterable) loader forever; every pass re-iterates it, so a shuffling loader reshuffles."""
    while True:
        yielded = False
        for batch in data_loader:
            yielded = True
            yield batch
        if not yielded:
            raise ValueError("infinite_dataloader: the loader is empty")
