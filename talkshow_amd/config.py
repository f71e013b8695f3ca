"""JSON config -> attribute tree with the reference's names (`trainer/config.py:10-22`: `Object`, `load_JsonConfig`)."""
import json


class Object:
    """Nested dicts become nested `Object`s, every other value an attribute (lists stay lists, as in the reference)."""

    def __init__(self, config: dict) -> None:
        for key, value in config.items():
            setattr(self, key, Object(value) if isinstance(value, dict) else value)

    def __repr__(self):
        return "Object(%r)" % (vars(self),)


def load_JsonConfig(json_file):
    with open(json_file) as f:
        return Object(json.load(f))
