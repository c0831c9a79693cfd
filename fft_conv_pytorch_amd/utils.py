"""Host-side helpers (row a1 of SURVEY section 8a)."""
from collections.abc import Iterable
from typing import Tuple, Union


def to_ntuple(val: Union[int, "Iterable[int]"], n: int) -> Tuple[int, ...]:
    """Expand an int to an ``n``-tuple; pass an iterable through after a length check.

    Same contract as the reference helper (/root/reference/fft_conv_pytorch/utils.py:4-20),
    including its error text and the fact that any iterable -- a ``str`` such as
    ``padding="same"`` too -- is expanded element-wise (so string padding ends in
    ``ValueError``, as it does in the reference).
    """
    if not isinstance(val, Iterable):
        return (val,) * n
    items = tuple(val)
    if len(items) != n:
        raise ValueError(f"Cannot cast tuple of length {len(items)} to length {n}.")
    return items
