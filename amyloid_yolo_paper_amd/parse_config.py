"""Darknet cfg / data-config readers.

Same file formats and return conventions as the reference's
``utils/parse_config.py:3-36`` (list of string-valued dicts, ``type`` key,
``batch_normalize`` defaulting to 0 on convolutional blocks; data config with
the ``gpus`` / ``num_workers`` defaults), written from the format description
in SURVEY.md App. C.1, so existing ``.cfg`` / ``.data`` files keep working.
"""


def parse_model_config(path):
    """cfg text -> list of blocks; every value stays a *string* (reference behaviour)."""
    blocks = []
    with open(path, "r") as fh:
        for raw in fh.read().split("\n"):
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            if line.startswith("["):
                block = {"type": line[1:-1].rstrip()}
                if block["type"] == "convolutional":
                    block["batch_normalize"] = 0
                blocks.append(block)
                continue
            key, value = line.split("=")
            blocks[-1][key.rstrip()] = value.strip()
    return blocks


def parse_data_config(path):
    """``key = value`` data config (classes/train/valid/names)."""
    options = {"gpus": "0,1,2,3", "num_workers": "10"}
    with open(path, "r") as fh:
        for raw in fh.readlines():
            line = raw.strip()
            if line == "" or line.startswith("#"):
                continue
            key, value = line.split("=")
            options[key.strip()] = value.strip()
    return options
