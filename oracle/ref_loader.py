"""Import the REAL reference (``/root/reference``) for golden-vector generation.

TEST INFRASTRUCTURE, build-container only: the reference does not exist on the GPU
box and never travels (no source, no bytecode).  The reference's package
``__init__`` files import torchvision / pytorch_msssim / pdqhash, which are absent
from this image, so the torch-only hot-path files are loaded by file path and the
``networks`` / ``tools`` packages are pre-seeded in ``sys.modules`` with exactly
those classes; ``models`` then imports unmodified.
"""
import importlib
import importlib.util
import os
import sys
import types

REF = os.environ.get('IPRGAN_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF, 'networks'))


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load():
    """Returns (networks, tools, models) namespaces of the real reference."""
    sys.dont_write_bytecode = True              # never write __pycache__ into /root/reference
    if 'ref_models_loaded' in sys.modules:
        m = sys.modules['ref_models_loaded']
        return m.networks, m.tools, m.models
    networks = types.ModuleType('networks')
    for f in ('conv_generator', 'sn_discriminator', 'conv_discriminator', 'sr_resnet',
              'discriminator_96', 'resnet_generator', 'encoder', 'decoder'):
        mod = _load(f'_ref_net_{f}', f'networks/{f}.py')
        for k, v in vars(mod).items():
            if not k.startswith('__'):
                setattr(networks, k, v)
    # BASELINE config 5 (128x128 DCGAN): the reference has the classes but no factory names; models.DCGAN looks
    # networks up by name (models/dcgan.py:10-11), so the two names are registered over the reference's OWN classes
    networks.ConvGenerator128 = lambda: networks.ConvGenerator(mg=16)
    networks.SNDiscriminator128 = lambda: networks.SNDiscriminator(md=16)
    # networks/vgg.py needs torchvision (absent): the architecture is restated in oracle/nets.py and
    # injected under the reference's name so that the REAL models.SRGAN can be constructed.
    from . import nets as _oracle_nets
    networks.VGG19Feature = _oracle_nets.VGG19Feature
    tools = types.ModuleType('tools')
    sm = _load('_ref_tools_sign_model', 'tools/sign_model.py')
    tools.SignLossModel = sm.SignLossModel
    tools.BitGenerator = sm.BitGenerator
    # black-box pieces: the torch-only transforms are the reference's own; the ones that import torchvision /
    # pytorch_msssim (absent) are the restatements of oracle/bbox.py, injected under the reference's names so
    # that the REAL models.BlackBoxWrapper choreography (models/wrappers.py:7-74) can run.
    for f, cls in (('transform_dist', 'TransformDist'), ('random_bitmask', 'RandomBitMask'),
                   ('transform_var', 'TransformVar')):
        setattr(tools, cls, getattr(_load(f'_ref_tools_{f}', f'tools/{f}.py'), cls))
    from . import bbox as _oracle_bbox
    for name in ('RandomNoisePatch', 'PasteWatermark', 'ssim', 'l1', 'mse'):
        setattr(tools, name, getattr(_oracle_bbox, name))
    saved = {k: sys.modules.get(k) for k in ('networks', 'tools', 'models')}
    sys.modules['networks'] = networks
    sys.modules['tools'] = tools
    sys.path.insert(0, REF)
    try:
        models = importlib.import_module('models')
        configs = _load('_ref_configs', 'configs/__init__.py')
    finally:
        sys.path.remove(REF)
    holder = types.ModuleType('ref_models_loaded')
    holder.networks, holder.tools, holder.models, holder.Config = networks, tools, models, configs.Config
    sys.modules['ref_models_loaded'] = holder
    return networks, tools, models


def load_loss():
    """The REAL ``tools/loss.py`` (``Loss``, ``l1``, ``mse``: tools/loss.py:10-20,72-76).  Its module-level imports of
    torchvision.transforms.Normalize and pytorch_msssim.SSIM / MS_SSIM (both absent from this image, neither used by
    ``Loss`` / ``l1`` / ``mse``) are satisfied by empty stand-in modules that exist only while the file is loaded."""
    sys.dont_write_bytecode = True
    stubs = {'torchvision': types.ModuleType('torchvision'), 'torchvision.transforms': types.ModuleType('torchvision.transforms'),
             'pytorch_msssim': types.ModuleType('pytorch_msssim')}
    stubs['torchvision'].transforms = stubs['torchvision.transforms']
    stubs['torchvision.transforms'].Normalize = object
    stubs['pytorch_msssim'].SSIM = stubs['pytorch_msssim'].MS_SSIM = object
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        return _load('_ref_tools_loss', 'tools/loss.py')
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def Config(entries):
    load()
    return sys.modules['ref_models_loaded'].Config(entries)
