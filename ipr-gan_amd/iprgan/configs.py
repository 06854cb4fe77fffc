"""YAML config object with the reference's access patterns (configs/__init__.py:4-44):
attribute and ``[]`` access, ``.get``, nested dicts become Config, ``to_dict`` / ``to_yaml``,
in-place mutation by the experiment code."""
import json

import yaml


class Config(object):
    @classmethod
    def parse(cls, fpath):
        with open(fpath, 'r') as f:
            return cls(yaml.safe_load(f))

    def __init__(self, entries):
        for key, value in entries.items():
            self.__dict__[key] = Config(value) if type(value) is dict else value

    def __getitem__(self, key):
        return self.__dict__[key]

    def __setitem__(self, key, value):
        self.__dict__[key] = value

    def get(self, key, default=None):
        return self.__dict__.get(key, default)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Config) else v) for k, v in self.__dict__.items()}

    def __str__(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True)

    def to_yaml(self):
        return yaml.safe_dump(self.to_dict())
