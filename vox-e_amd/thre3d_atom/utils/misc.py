"""Small host utilities (API of the reference's thre3d_atom/utils/misc.py; none is on the GPU hot path)."""
import math
from pathlib import Path
from typing import Any, Callable, Iterator, List, Mapping, Optional, Sequence, Tuple

import yaml

GridSize = Tuple[int, int, int]


def check_power_of_2(x: int) -> bool:
    return (x & (x - 1)) == 0


class _Chunked:
    """callable that feeds the first argument of `fn` through in slices and collates the partial results"""

    def __init__(self, fn: Callable[..., Any], collate: Callable[[Sequence[Any]], Any], size: int, verbose: bool):
        self.fn, self.collate, self.size, self.verbose = fn, collate, size, verbose

    def _slices(self, total: int) -> Iterator[slice]:
        starts = range(0, total, self.size)
        if self.verbose:
            from tqdm import tqdm

            starts = tqdm(starts)
        return (slice(s, s + self.size) for s in starts)

    def __call__(self, inputs, *args, **kwargs):
        return self.collate([self.fn(inputs[sl], *args, **kwargs) for sl in self._slices(len(inputs))])


def batchify(processor_fn: Callable[..., Any], collate_fn: Callable[[Sequence[Any]], Any],
             chunk_size: Optional[int] = None, verbose: bool = False) -> Callable[..., Any]:
    """`processor_fn` itself when chunk_size is None, else a chunk-wise wrapper (reference misc.py:14-35).
    The fused renderer never needs it (no per-sample temporaries); kept for API parity."""
    return processor_fn if chunk_size is None else _Chunked(processor_fn, collate_fn, chunk_size, verbose)


def compute_thre3d_grid_sizes(final_required_resolution: GridSize, num_stages: int, scale_factor: float) -> List[GridSize]:
    """coarse-to-fine schedule, finest last: each earlier stage is ceil(size / scale_factor) per axis
    (20 -> 40 -> 80 -> 160 for 160^3, 4 stages, factor 2; reference misc.py:38-50)"""
    schedule = [tuple(int(v) for v in final_required_resolution)]
    while len(schedule) < num_stages:
        schedule.insert(0, tuple(int(math.ceil((1 / scale_factor) * v)) for v in schedule[0]))
    return schedule


def log_config_to_disk(args: Mapping[str, Any], output_dir: Path, config_file_name: str = "config.yml") -> None:
    target = Path(output_dir)
    target.mkdir(exist_ok=True, parents=True)
    (target / config_file_name).write_text(yaml.dump(dict(args), default_flow_style=False))
