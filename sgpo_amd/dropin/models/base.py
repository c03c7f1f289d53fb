"""Mirror of the reference's `GenerativeModel` duck type (models/base.py:4-18)."""
from abc import ABC


class GenerativeModel(ABC):
    def __init__(self):
        pass

    def score(self, x, t):
        pass

    def pred_mean(self, x, t):
        pass

    def get_start(self):
        pass

    def q_sample(self, x, t):
        pass


def cfg_get(cfg, path, default=None):
    """Read `a.b.c` from a dict / hydra DictConfig / attribute object."""
    cur = cfg
    for part in path.split('.'):
        if cur is None:
            return default
        if isinstance(cur, dict):
            cur = cur.get(part, None)
        else:
            try:
                cur = getattr(cur, part)
            except (AttributeError, KeyError):
                try:
                    cur = cur[part]
                except Exception:
                    return default
    return default if cur is None else cur
