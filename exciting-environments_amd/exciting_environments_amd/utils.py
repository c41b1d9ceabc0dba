"""Min-max normalisation record and the JSON format of the golden fixtures' ``sim_properties.json``
(behaviour of reference exciting_environments/utils.py:8-52)."""
import dataclasses
import json


@dataclasses.dataclass
class MinMaxNormalization:
    """Maps [min, max] to [-1, 1] and back. ``min`` / ``max`` are Python scalars or [batch_size] arrays (torch / numpy).
    The operation order is part of the contract (the kernels reproduce it exactly): utils.py:13-17."""

    min: float
    max: float

    def normalize(self, denormalized_value):
        span = self.max - self.min
        return 2 * (denormalized_value - self.min) / span - 1

    def denormalize(self, normalized_value):
        span = self.max - self.min
        return (normalized_value + 1) / 2 * span + self.min


def _bounds_to_plain(norms: dict) -> dict:
    return {name: {"min": n.min, "max": n.max} for name, n in norms.items()}


def _bounds_from_plain(plain: dict) -> dict:
    return {name: MinMaxNormalization(min=b["min"], max=b["max"]) for name, b in plain.items()}


def dump_sim_properties_to_json(params, action_normalizations, physical_normalizations, tau, filename):
    """Write {params, action_normalizations, physical_normalizations, tau} in the fixture format."""
    payload = {
        "params": params,
        "action_normalizations": _bounds_to_plain(action_normalizations),
        "physical_normalizations": _bounds_to_plain(physical_normalizations),
        "tau": tau,
    }
    with open(filename, "w") as fh:
        json.dump(payload, fh, indent=4)


def load_sim_properties_from_json(filename):
    """Inverse of dump_sim_properties_to_json: (params, action_normalizations, physical_normalizations, tau)."""
    with open(filename) as fh:
        payload = json.load(fh)
    return (payload["params"], _bounds_from_plain(payload["action_normalizations"]),
            _bounds_from_plain(payload["physical_normalizations"]), payload["tau"])
