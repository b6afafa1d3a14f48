"""Import the avex reference in THIS container only (tooling for golden generation).

/root/reference does not exist on the GPU box; nothing under tests/ that runs there
imports this module. Several optional dependencies of the reference are absent in
this image, so inert placeholder modules are registered before ``import avex``
(they are never called on the BEATs path).
"""
import sys
sys.dont_write_bytecode = True
import types, importlib.machinery, importlib.metadata as ilm
from pydantic import BaseModel

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
    return m


def import_reference():
    if "avex" in sys.modules:
        return sys.modules["avex"]

    class BaseSettings(BaseModel):
        def __init_subclass__(cls, **kw):
            kw.pop("extra", None)
            kw.pop("validate_assignment", None)
            super().__init_subclass__()

    class _S:
        def __init__(self, *a, **k):
            pass

    sys.path.insert(0, REF)
    _stub("pydantic_settings", BaseSettings=BaseSettings, CliSettingsSource=_S, YamlConfigSettingsSource=_S)
    _stub("gcsfs", GCSFileSystem=type("GCSFileSystem", (), {}))
    _stub("s3fs", S3FileSystem=type("S3FileSystem", (), {}))
    _stub("h5py")
    ta = _stub("torchaudio")
    ta.transforms = _stub("torchaudio.transforms")
    ta.functional = _stub("torchaudio.functional")
    ta.compliance = _stub("torchaudio.compliance")
    ta.compliance.kaldi = _stub("torchaudio.compliance.kaldi")
    _v = ilm.version
    ilm.version = lambda n: "1.2.0" if n == "avex" else _v(n)
    import avex  # noqa: F401
    return avex
