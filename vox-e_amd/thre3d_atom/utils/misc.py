"""Small host utilities (reference: thre3d_atom/utils/misc.py)."""
import math
from pathlib import Path
from typing import Any, Callable, List, Mapping, Optional, Sequence, Tuple

import yaml


def check_power_of_2(x: int) -> bool:
    return x & (x - 1) == 0


def batchify(
    processor_fn: Callable[..., Any],
    collate_fn: Callable[[Sequence[Any]], Any],
    chunk_size: Optional[int] = None,
    verbose: bool = False,
) -> Callable[..., Any]:
    """Wrap `processor_fn` so that its first argument is processed in chunks (misc.py:14-35).
    The fused renderer never needs this (it has no per-sample temporaries); kept for API parity."""
    if chunk_size is None:
        return processor_fn

    def chunked(inputs, *args, **kwargs):
        starts = range(0, len(inputs), chunk_size)
        if verbose:
            from tqdm import tqdm

            starts = tqdm(starts)
        return collate_fn([processor_fn(inputs[s: s + chunk_size], *args, **kwargs) for s in starts])

    return chunked


def compute_thre3d_grid_sizes(
    final_required_resolution: Tuple[int, int, int], num_stages: int, scale_factor: float
) -> List[Tuple[int, int, int]]:
    """Coarse-to-fine grid schedule: ceil(size / scale_factor) per earlier stage (misc.py:38-50)."""
    sizes = [tuple(int(v) for v in final_required_resolution)]
    for _ in range(num_stages - 1):
        sizes.insert(0, tuple(int(math.ceil((1 / scale_factor) * v)) for v in sizes[0]))
    return sizes


def log_config_to_disk(args: Mapping[str, Any], output_dir: Path, config_file_name: str = "config.yml") -> None:
    output_dir = Path(output_dir)
    output_dir.mkdir(exist_ok=True, parents=True)
    with open(output_dir / config_file_name, "w") as fh:
        yaml.dump(dict(args), fh, default_flow_style=False)
