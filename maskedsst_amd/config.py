"""Config loading with the reference's semantics (``src/utils.py:316-334``): the pretrain YAML is
overlaid with ``data[dataset]``, ``transformer`` and ``masked_modeling`` of the general YAML into one
flat attribute bag."""
import yaml


class Dotdict(object):
    def __init__(self, data):
        self.__dict__.update(data)

    def __repr__(self):
        return f"Dotdict({self.__dict__})"


def get_pretrain_config(pretrain_config_path, general_config_path, seed, device):
    with open(pretrain_config_path, "r") as f:
        hp = yaml.safe_load(f)
    with open(general_config_path, "r") as f:
        general = yaml.safe_load(f)
    hp.update(general["data"][hp["dataset"]])
    hp.update(general["transformer"])
    hp.update(general["masked_modeling"])
    hp["seed"] = seed
    hp["device"] = device
    return Dotdict(hp)
