"""Linear-feature (trail) detection on SDSS-like frames, MI355X-native.

Same public names as ``lfd.detecttrails`` (lfd/detecttrails/__init__.py:69-71):
``DetectTrails``, ``process_field``, ``process_field_bright``, ``process_field_dim``,
``remove_stars``, ``read_photoObj``, plus ``setup`` for the data-tree environment variables.
"""
import os as _os


def setup(bosspath=None, photoobjpath=None, photoreduxpath=None, debugpath=None):
    """Set $BOSS, $BOSS_PHOTOOBJ, $PHOTO_REDUX and $DEBUG_PATH (reference:
    lfd/detecttrails/__init__.py:36-66; defaults ~/Desktop/boss, $BOSS/photoObj, $BOSS/photo/redux, ~/Desktop/debug)."""
    boss = _os.path.expanduser(bosspath) if bosspath else _os.path.expanduser("~/Desktop/boss")
    _os.environ["BOSS"] = boss
    _os.environ["BOSS_PHOTOOBJ"] = photoobjpath or _os.path.join(boss, "photoObj")
    _os.environ["PHOTO_REDUX"] = photoreduxpath or _os.path.join(boss, "photo", "redux")
    _os.environ["DEBUG_PATH"] = _os.path.expanduser(debugpath) if debugpath else _os.path.expanduser("~/Desktop/debug")


from .removestars import *  # noqa: E402,F401,F403
from .processfield import *  # noqa: E402,F401,F403
from .detecttrails import *  # noqa: E402,F401,F403
from .detecttrails import (RETR_LIST, RETR_EXTERNAL, RETR_CCOMP, RETR_TREE,  # noqa: E402,F401
                           CHAIN_APPROX_NONE, CHAIN_APPROX_SIMPLE, CHAIN_APPROX_TC89_L1,
                           CHAIN_APPROX_TC89_KCOS)
