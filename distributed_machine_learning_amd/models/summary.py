"""Per-layer table in the layout the reference prints through ``torchsummary``
(/root/reference/part1/main.py:118, part2/2a/main.py:179, part3/main.py:135).

The table is computed from the layer plan; no forward pass and no torchsummary
dependency.  Column widths, rulers and the four size lines follow the table
reproduced on page 1 of the reference report (9,225,610 params, 35.19 MB params,
37.76 MB estimated total for the no-BN VGG-11).
"""
from __future__ import annotations

from typing import List, Tuple

from .vgg import IMAGE_HW, IN_CHANNELS, layer_plan

_TYPE_NAME = {"conv": "Conv2d", "bn": "BatchNorm2d", "relu": "ReLU",
              "pool": "MaxPool2d", "fc": "Linear"}
_MB = 1024.0 ** 2


def summary_rows(arch: str = "VGG11", batch_norm: bool = False,
                 num_classes: int = 10) -> List[Tuple[str, List[int], int]]:
    """(name, output shape with -1 batch, parameter count) per leaf module."""
    rows = []
    for n, s in enumerate(layer_plan(arch, batch_norm, num_classes), start=1):
        name = f"{_TYPE_NAME[s.kind]}-{n}"
        if s.kind == "conv":
            shape, params = [-1, s.cout, s.hw, s.hw], s.cout * s.cin * 9 + s.cout
        elif s.kind == "bn":
            shape, params = [-1, s.cout, s.hw, s.hw], 2 * s.cout
        elif s.kind == "relu":
            shape, params = [-1, s.cout, s.hw, s.hw], 0
        elif s.kind == "pool":
            shape, params = [-1, s.cout, s.hw // 2, s.hw // 2], 0
        else:
            shape, params = [-1, s.cout], s.cin * s.cout + s.cout
        rows.append((name, shape, params))
    return rows


def summary_string(arch: str = "VGG11", batch_norm: bool = False,
                   input_size=(IN_CHANNELS, IMAGE_HW, IMAGE_HW), num_classes: int = 10) -> str:
    rows = summary_rows(arch, batch_norm, num_classes)
    ruler, double = "-" * 64, "=" * 64
    fmt = "{:>20}  {:>25} {:>15}"
    out = [ruler, fmt.format("Layer (type)", "Output Shape", "Param #"), double]
    total_params = 0
    total_out = 0
    for name, shape, params in rows:
        out.append(fmt.format(name, str(shape), "{0:,}".format(params)))
        total_params += params
        elems = 1
        for d in shape[1:]:
            elems *= d
        total_out += elems
    in_elems = 1
    for d in input_size:
        in_elems *= d
    input_mb = in_elems * 4.0 / _MB
    fwd_bwd_mb = 2.0 * total_out * 4.0 / _MB
    params_mb = total_params * 4.0 / _MB
    out += [
        double,
        "Total params: {0:,}".format(total_params),
        "Trainable params: {0:,}".format(total_params),
        "Non-trainable params: {0:,}".format(0),
        ruler,
        "Input size (MB): %0.2f" % input_mb,
        "Forward/backward pass size (MB): %0.2f" % fwd_bwd_mb,
        "Params size (MB): %0.2f" % params_mb,
        "Estimated Total Size (MB): %0.2f" % (input_mb + fwd_bwd_mb + params_mb),
        ruler,
    ]
    return "\n".join(out)


def print_summary(arch: str = "VGG11", batch_norm: bool = False, echo_none: bool = True) -> None:
    """Print the table.  The reference wraps the call in ``print(...)`` so a literal
    ``None`` line follows the table in its logs; ``echo_none`` keeps that line."""
    print(summary_string(arch, batch_norm))
    if echo_none:
        print(None)
