"""``flatten_dict`` of reference ``nr4seg/utils/flatten_dict.py``: nested
mappings become one level, keys joined with ``sep``; used to log the
experiment configuration."""
from collections.abc import Mapping

__all__ = ["flatten_dict"]


def _leaves(node, prefix, sep):
    """(joined key, value) of every non-mapping value, in insertion order."""
    for key, value in node.items():
        name = f"{prefix}{sep}{key}" if prefix else key
        if isinstance(value, Mapping):
            yield from _leaves(value, name, sep)
        elif isinstance(value, list) and value and isinstance(value[0], Mapping):
            # lists of mappings: the position becomes part of the key
            for pos, item in enumerate(value):
                yield from _leaves(item, f"{name}{sep}{pos}", sep)
        else:
            yield name, value


def flatten_dict(d, parent_key="", sep="_"):
    return dict(_leaves(d, parent_key, sep))
