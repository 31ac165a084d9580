"""Import shim: the product package lives in ``stark-symphony_amd/`` (the directory name
the project layout prescribes, which Python cannot import because of the hyphen).  This
package only redirects its search path there, so ``import stark_symphony_amd.formats``
loads ``stark-symphony_amd/formats.py``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "stark-symphony_amd")
__path__.insert(0, _real)

from ._exports import *  # noqa: E402,F401,F403
from ._exports import __all__  # noqa: E402,F401
