"""Normalisation helper and fixture I/O — mirrors reference exciting_environments/utils.py."""
from dataclasses import asdict, dataclass
import json


@dataclass
class MinMaxNormalization:
    """utils.py:8-17. `min`/`max` may be Python scalars or [batch_size] arrays (torch / numpy)."""

    min: float
    max: float

    def normalize(self, denormalized_value):
        return 2 * (denormalized_value - self.min) / (self.max - self.min) - 1

    def denormalize(self, normalized_value):
        return (normalized_value + 1) / 2 * (self.max - self.min) + self.min


def dump_sim_properties_to_json(params, action_normalizations, physical_normalizations, tau, filename):
    """utils.py:21-35 (format of the golden fixtures' sim_properties.json)."""
    data = {
        "params": params,
        "action_normalizations": {k: asdict(v) for k, v in action_normalizations.items()},
        "physical_normalizations": {k: asdict(v) for k, v in physical_normalizations.items()},
        "tau": tau,
    }
    with open(filename, "w") as f:
        json.dump(data, f, indent=4)


def load_sim_properties_from_json(filename):
    """utils.py:37-52."""
    with open(filename, "r") as f:
        data = json.load(f)
    action_normalizations = {k: MinMaxNormalization(**v) for k, v in data["action_normalizations"].items()}
    physical_normalizations = {k: MinMaxNormalization(**v) for k, v in data["physical_normalizations"].items()}
    return data["params"], action_normalizations, physical_normalizations, data["tau"]
