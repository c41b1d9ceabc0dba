"""Minimal pytree helpers over the State / EnvProperties dataclasses (stand-in for jax.tree_util in the
reference's tests: tree_flatten / tree_structure of states, tests/envs/test_core_functions.py:85-100)."""
from dataclasses import fields, is_dataclass


def tree_flatten(x):
    leaves = []
    struct = _flatten(x, leaves)
    return leaves, struct


def _flatten(x, leaves):
    if x is None:
        return None
    if is_dataclass(x) and not isinstance(x, type):
        return (type(x).__qualname__, tuple((f.name, _flatten(getattr(x, f.name), leaves)) for f in fields(x)))
    if isinstance(x, (tuple, list)):
        return (type(x).__name__, tuple(_flatten(v, leaves) for v in x))
    leaves.append(x)
    return "*"


def tree_structure(x):
    return tree_flatten(x)[1]


def tree_map(fn, x):
    if x is None:
        return None
    if is_dataclass(x) and not isinstance(x, type):
        return type(x)(**{f.name: tree_map(fn, getattr(x, f.name)) for f in fields(x)})
    if isinstance(x, (tuple, list)):
        return type(x)(tree_map(fn, v) for v in x)
    return fn(x)
