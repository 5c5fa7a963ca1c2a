"""Import shim: the product package lives in the directory `whisper-rust-ort_amd/` (a name
Python cannot import directly); this module makes it importable as `whisper_rust_ort_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "whisper-rust-ort_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
