"""Logger and yaml helpers (host-side mirror of reference yolov6/utils/events.py:9-30)."""
import logging
import os
import shutil

import yaml


def set_logging(name=None):
    rank = int(os.getenv('RANK', -1))
    logging.basicConfig(format="%(message)s", level=logging.INFO if rank in (-1, 0) else logging.WARNING)
    return logging.getLogger(name)


LOGGER = set_logging(__name__)
NCOLS = min(200, shutil.get_terminal_size().columns)


def load_yaml(file_path):
    if isinstance(file_path, str):
        with open(file_path, errors='ignore') as f:
            return yaml.safe_load(f)


def save_yaml(data_dict, save_path):
    with open(save_path, 'w') as f:
        yaml.safe_dump(data_dict, f, sort_keys=False)
