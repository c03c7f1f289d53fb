"""Drop-in `dataset`: the modules mirrored here win; any other `dataset.*` falls through to the same-named directory
of the reference code base when that is on sys.path (sgpo_amd/dropin/_fallthrough.py)."""
from sgpo_amd.dropin._fallthrough import FallthroughPath

__path__ = FallthroughPath(__path__, __name__)
