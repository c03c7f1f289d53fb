"""Import fall-through for the drop-in packages.

The drop-in mirrors only the modules on the sampling path (`sampling.cls_guidance`, `models.mdlm`, ...).  The
reference's drivers also import modules that are NOT on that path (`sampling.DPO`, `training.train_GP_classifier`,
`models.continuous`, `models.causalLM`, `models.pretraining.model.*`: prior_sample.py:14-15, iterativeBO.py:17,
sampling/uncond.py:5-8).  A regular package shadows a same-named directory later on `sys.path`, so without help those
imports fail with the drop-in installed.  Each drop-in package therefore sets

    __path__ = FallthroughPath(__path__, __name__)

which lists the package's own directory first (mirrored modules win) and then every same-named directory found under
the later entries of `sys.path` -- the reference root when a driver runs from there.  The list follows `sys.path`:
it is recomputed whenever `sys.path` has changed, like the path of a namespace package.
"""
import os
import sys


class FallthroughPath:
    def __init__(self, own, name):
        self._own = [os.path.abspath(p) for p in own]
        self._rel = os.path.join(*name.split('.'))
        self._seen = None
        self._cache = list(self._own)

    def _current(self):
        key = tuple(sys.path)
        if key != self._seen:
            out = list(self._own)
            dropin = os.path.dirname(os.path.abspath(__file__))
            for entry in sys.path:
                root = os.path.abspath(entry or os.getcwd())
                if root == dropin:
                    continue
                cand = os.path.join(root, self._rel)
                if os.path.isdir(cand) and cand not in out:
                    out.append(cand)
            self._seen, self._cache = key, out
        return self._cache

    def __iter__(self):
        return iter(self._current())

    def __len__(self):
        return len(self._current())

    def __getitem__(self, i):
        return self._current()[i]

    def __contains__(self, item):
        return item in self._current()

    def __repr__(self):
        return f'FallthroughPath({self._current()!r})'

    def append(self, item):
        self._own.append(os.path.abspath(item))
        self._seen = None

    def fallthrough_dirs(self):
        """The same-named directories of the other code bases on sys.path (without the drop-in's own)."""
        return [p for p in self._current() if p not in self._own]


def reference_file(package, filename):
    """Path of `filename` in the first fall-through directory of `package` that has it (None if there is none)."""
    path = getattr(sys.modules.get(package), '__path__', None)
    if not isinstance(path, FallthroughPath):
        return None
    for d in path.fallthrough_dirs():
        f = os.path.join(d, filename)
        if os.path.isfile(f):
            return f
    return None


def load_reference_module(package, stem):
    """Import the reference's OWN `package/stem.py` under the private name `package._reference_<stem>` (the public
    name belongs to the drop-in mirror); relative imports inside it resolve within `package`, i.e. to the mirrors."""
    import importlib.util
    name = f'{package}._reference_{stem}'
    if name in sys.modules:
        return sys.modules[name]
    f = reference_file(package, stem + '.py')
    if f is None:
        raise ImportError(f'{package}.{stem}: the reference code base is not on sys.path (needed for a name the '
                          f'MI355X drop-in does not mirror)')
    spec = importlib.util.spec_from_file_location(name, f)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = package
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(name, None)
        raise
    return mod


def forward_missing_names(package, stem):
    """Module-level `__getattr__` for a mirror that covers only part of its reference module: a name the mirror does
    not define (e.g. `sampling.cls_guidance.Classifier_Guidance_Continuous`, prior_sample.py:15) is looked up in the
    reference's own file, imported lazily on first use."""
    def __getattr__(name):
        if name.startswith('__'):
            raise AttributeError(name)
        try:
            ref = load_reference_module(package, stem)
        except ImportError as e:
            raise AttributeError(f'module {package}.{stem!r} has no attribute {name!r} ({e})') from e
        try:
            return getattr(ref, name)
        except AttributeError:
            raise AttributeError(f'module {package}.{stem!r} has no attribute {name!r}') from None
    return __getattr__
