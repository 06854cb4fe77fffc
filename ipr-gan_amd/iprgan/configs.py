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


# ---- engine block -------------------------------------------------------------------------------------------------
# The reference's YAML has no notion of an engine; ours adds one optional top-level block that the train driver applies
# before it builds the model (every key is optional, the defaults are the environment-variable defaults):
#
#   engine:
#     math: bf16act          # fp32 | fp32x3 | bf16 | bf16act           (iprgan_set_math_mode, include/iprgan.h)
#     bucket_mb: 8           # gradient bucket size of the N > 1 reducer           (IPRGAN_BUCKET_MB)
#     comm: rccl             # rccl (C-ABI communicator) | torch (torch.distributed) (IPRGAN_COMM)
#     comm_timeout: 90       # seconds allowed for the RCCL bring-up               (IPRGAN_COMM_TIMEOUT)
#     tune_cache: path.json  # autotune choices replayed / stored across runs      (IPRGAN_TUNE_CACHE)
#     fuse_stats: true       # norm statistics / bias gradients from conv epilogues (IPRGAN_FUSE_STATS)
#     pair_d: true           # D(real) and D(fake) as one paired pass              (IPRGAN_PAIR_D)
#     batch_passes: true     # same-network passes of CycleGAN batched             (IPRGAN_BATCH_PASSES)
#     graph: true            # ImageGeneration: the whole step as one captured HIP graph (graphs.py); one rank only, unless
#     graph_ddp: true        #   ... this opts a data-parallel run in (the exchange through the C ABI's communicator is captured
#                            #   with the step; every rank captures or every rank stays eager; unproven on real multi-GPU RCCL)
ENGINE_KEYS = ('math', 'bucket_mb', 'comm', 'comm_timeout', 'tune_cache', 'fuse_stats', 'pair_d', 'batch_passes', 'graph',
               'graph_ddp')


def apply_engine(config, set_math=True):
    """Apply the ``engine:`` block of a parsed config (a Config, a dict or None).  Returns the dict of settings that were
    applied.  Unknown keys are an error: a typo must not silently train in the wrong mode."""
    import os
    blk = config.get('engine', None) if config is not None else None
    if blk is None:
        return {}
    blk = blk.to_dict() if isinstance(blk, Config) else dict(blk)
    unknown = sorted(set(blk) - set(ENGINE_KEYS))
    if unknown:
        raise ValueError(f'engine: unknown key(s) {unknown}; known: {list(ENGINE_KEYS)}')
    if 'math' in blk:
        if blk['math'] not in ('fp32', 'bf16', 'bf16act', 'fp32x3'):
            raise ValueError(f"engine.math: {blk['math']!r} (fp32 | bf16 | bf16act | fp32x3)")
        if set_math:
            from . import _lib
            _lib.set_math(blk['math'])
    if 'comm' in blk and blk['comm'] not in ('rccl', 'torch'):
        raise ValueError(f"engine.comm: {blk['comm']!r} (rccl | torch)")
    for key, env in (('bucket_mb', 'IPRGAN_BUCKET_MB'), ('comm', 'IPRGAN_COMM'), ('comm_timeout', 'IPRGAN_COMM_TIMEOUT'),
                     ('tune_cache', 'IPRGAN_TUNE_CACHE')):
        if key in blk:
            os.environ[env] = str(blk[key])
    # switches that modules read once at import: set the module attributes as well as the environment
    from . import engine as _engine, models as _models
    for key, env, mod, attr in (('fuse_stats', 'IPRGAN_FUSE_STATS', _engine, '_FUSE_STATS'),
                                ('pair_d', 'IPRGAN_PAIR_D', _models, '_PAIR_D'),
                                ('batch_passes', 'IPRGAN_BATCH_PASSES', _models, '_BATCH_PASSES')):
        if key in blk:
            os.environ[env] = '1' if blk[key] else '0'
            setattr(mod, attr, bool(blk[key]))
    return blk
