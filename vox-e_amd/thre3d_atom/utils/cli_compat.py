"""Command-line compatibility with the reference's entry points (new in this build).

The reference's scripts (train_sh_based_voxel_grid_with_posed_images.py, edit_pretrained_relu_field.py,
refine_edited_relu_field.py) declare several options that the reference itself never reads (grid geometry on scripts
that load a trained model, wandb account names, data-loader workers ...) or that configure machinery this build does not
have.  `accepted_options` declares them on a click command with the reference's names, types and defaults, so the
reference's shell scripts run unchanged; `report_unused` logs which of them were given a non-default value."""
from typing import Any, Callable, Dict, Sequence, Tuple

import click

# (name, click type, default, nargs)
Spec = Tuple[str, Any, Any, int]


def accepted_options(specs: Sequence[Spec]) -> Callable:
    def decorate(fn: Callable) -> Callable:
        for name, typ, default, nargs in reversed(list(specs)):
            kw: Dict[str, Any] = dict(type=typ, default=default, show_default=True, required=False,
                                      help="accepted for command-line compatibility with the reference")
            if nargs and nargs > 1:
                kw["nargs"] = nargs
            fn = click.option(name, **kw)(fn)
        return fn

    return decorate


def report_unused(kwargs: Dict[str, Any], specs: Sequence[Spec], log) -> None:
    changed = []
    for name, _typ, default, _nargs in specs:
        key = name.lstrip("-")
        if key not in kwargs:
            continue
        value = kwargs[key]
        same = tuple(value) == tuple(default) if isinstance(default, (tuple, list)) else value == default
        if not same:
            changed.append(f"{name}={value}")
    if changed:
        log.info("options accepted for compatibility with the reference and not used by this build: " + ", ".join(changed))
