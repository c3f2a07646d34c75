"""Import alias for the package directory ``lina-speech_amd/``.

The product package lives in a directory whose name carries a hyphen (the
name the build contract fixes), which Python cannot import directly.  This
module turns itself into a package whose search path is that directory, so
``import lina_speech_amd.ops`` resolves to ``lina-speech_amd/ops.py``.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lina-speech_amd")]
__version__ = "0.1.0"
