"""reference nr4seg/utils/loading.py:14-17."""
import yaml


def load_yaml(path):
    with open(path) as f:
        return yaml.safe_load(f)
