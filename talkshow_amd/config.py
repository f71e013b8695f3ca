"""JSON -> attribute tree, as the reference's `trainer/config.py:10-22` (`Object`, `load_JsonConfig`)."""
import json


class Object():
    def __init__(self, config: dict) -> None:
        for key in list(config.keys()):
            if isinstance(config[key], dict):
                setattr(self, key, Object(config[key]))
            else:
                setattr(self, key, config[key])


def load_JsonConfig(json_file):
    with open(json_file, 'r') as f:
        config = json.load(f)
    return Object(config)
