"""Import the read-only reference (sippy/Infernos) in THIS container only.

Used solely by tools/gen_golden.py to capture golden vectors. Nothing here (nor the
reference) travels to the GPU box; only the numbers written to tests/golden/ do.

The reference imports several third-party packages that are absent from the image
(torchaudio, ctranslate2, ray, soundfile, methodtools, argostranslate).  They are not
needed for the arithmetic captured here, so empty module objects are registered for
them.  No reference source is copied anywhere.
"""
import sys
import types
import functools
from importlib.machinery import ModuleSpec

REF_ROOT = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    sys.dont_write_bytecode = True
    import transformers  # noqa: F401  (must be imported before the stubs exist)
    _stub('soundfile')
    ta = _stub('torchaudio')
    ta.transforms = _stub('torchaudio.transforms', Resample=None)
    ag = _stub('argostranslate')
    ag.package = _stub('argostranslate.package')
    ag.translate = _stub('argostranslate.translate', get_installed_languages=lambda: [])
    _stub('methodtools', lru_cache=functools.lru_cache)
    ct2 = _stub('ctranslate2')
    ct2.models = types.SimpleNamespace(Whisper=object)
    ct2.StorageView = object

    class _Remote:
        def __call__(self, *a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda c: c
    _stub('ray', remote=_Remote(), get=lambda x: x)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
