"""CPU, build container only (needs the reference tree): with the drop-in installed ahead of the reference root --
the order `python -m sgpo_amd.run <driver>` sets up -- every first-party import of the reference's drivers and every
first-party `_target_` of its hydra configs resolves: mirrored modules to the drop-in, the rest by falling through to
the reference (sgpo_amd/dropin/_fallthrough.py).  Third-party packages this image lacks are replaced by empty
stand-ins for the duration of the check, which runs in a child process so that no stub leaks into other tests."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
REF = '/root/reference'

CHECK = r'''
import ast, importlib, importlib.abc, importlib.machinery, json, os, sys, types
ROOT, REF = sys.argv[1], sys.argv[2]
FIRST_PARTY = {'models', 'sampling', 'problem', 'dataset', 'training', 'util'}
ABSENT_OK = {'hydra', 'omegaconf', 'wandb', 'Bio', 'botorch', 'gpytorch', 'evodiff', 'sequence_models', 'flash_attn',
             'pytorch_lightning', 'lightning', 'torchmetrics', 'timm', 'esm'}
# submodules that the image's newer release of an installed package no longer has (models/pretraining/model/progen2)
ABSENT_SUBMODULES_OK = {'transformers.utils.model_parallel_utils'}


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        if name == 'main':                      # hydra.main(...) is used as a decorator factory
            return lambda *a, **k: (lambda f: f)
        cls = type(name, (), {'__init__': lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.split('.')[0] in ABSENT_OK or name in ABSENT_SUBMODULES_OK:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


sys.meta_path.append(_Finder())                 # consulted last: only for packages that really are absent
sys.path.insert(0, ROOT)
import sgpo_amd
dropin = sgpo_amd.install_dropin()
sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (dropin, REF)]
sys.path[0:0] = [dropin, REF]                   # what sgpo_amd/run.py does for a driver under REF

report = {'imports': [], 'targets': [], 'skipped_targets': []}


def origin(obj, holder=None):
    mod = sys.modules.get(getattr(obj, '__module__', None) or '') or holder     # plain data (a dict): its module's file
    f = getattr(mod, '__file__', None) or getattr(obj, '__file__', '') or ''
    return 'dropin' if f.startswith(dropin) else ('reference' if f.startswith(REF) else f)


for driver in ('prior_sample.py', 'pareto.py', 'iterativeBO.py'):
    tree = ast.parse(open(os.path.join(REF, driver)).read())
    for node in tree.body:
        if isinstance(node, ast.ImportFrom) and node.module and node.module.split('.')[0] in FIRST_PARTY:
            mod = importlib.import_module(node.module)
            for a in node.names:
                obj = getattr(mod, a.name)
                report['imports'].append([driver, node.module, a.name, origin(obj, mod)])
        elif isinstance(node, ast.Import):
            for a in node.names:
                if a.name.split('.')[0] in FIRST_PARTY:
                    importlib.import_module(a.name)
                    report['imports'].append([driver, a.name, '', origin(sys.modules[a.name])])

import yaml


def targets(node, out):
    if isinstance(node, dict):
        for k, v in node.items():
            if k == '_target_' and isinstance(v, str):
                out.append(v)
            else:
                targets(v, out)
    elif isinstance(node, list):
        for v in node:
            targets(v, out)


cfg_root = os.path.join(REF, 'configs')
files = [os.path.join(cfg_root, f) for f in sorted(os.listdir(cfg_root)) if f.endswith('.yaml')]
for grp in ('algorithm', 'model', 'problem', 'data'):
    d = os.path.join(cfg_root, grp)
    files += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith('.yaml')]
for f in files:
    found = []
    targets(yaml.safe_load(open(f)), found)
    for t in found:
        modname, _, attr = t.rpartition('.')
        if modname.split('.')[0] not in FIRST_PARTY:
            continue
        rel = os.path.relpath(f, cfg_root)
        if not os.path.isfile(os.path.join(REF, *modname.split('.')) + '.py'):
            report['skipped_targets'].append([rel, t, 'no such module in the reference itself'])
            continue
        try:
            obj = getattr(importlib.import_module(modname), attr)
        except ModuleNotFoundError as e:
            # the reference's own module imports a first-party module that exists nowhere in the reference
            # (SURVEY.md defect D5: training/train_classifier_NOS.py:14 -> models.MDLM.*): not the boundary's to fix
            missing = (e.name or '').split('.')
            if missing[0] in FIRST_PARTY and not (os.path.exists(os.path.join(REF, *missing)) or
                                                  os.path.exists(os.path.join(REF, *missing) + '.py')):
                report['skipped_targets'].append([rel, t, f'the reference module imports {e.name}, which the reference lacks'])
                continue
            raise
        report['targets'].append([rel, t, origin(obj)])
print('REPORT ' + json.dumps(report))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference tree is not on this machine')
def test_driver_imports_and_config_targets_resolve_with_dropin_first(tmp_path):
    script = tmp_path / 'check.py'
    script.write_text(CHECK)
    env = dict(os.environ, PYTHONPATH='')
    r = subprocess.run([sys.executable, str(script), ROOT, REF], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rep = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('REPORT ')][-1][7:])
    imp = {(d, m, n): o for d, m, n, o in rep['imports']}
    # the names VERDICT round 1 found unresolvable
    assert imp[('prior_sample.py', 'sampling.DPO', 'DPOInpaint')] == 'reference'
    assert imp[('prior_sample.py', 'sampling.cls_guidance', 'Classifier_Guidance_Continuous')] == 'reference'
    assert imp[('prior_sample.py', 'sampling.cls_guidance', 'Classifier_Guidance_Inpaint')] == 'dropin'
    assert imp[('iterativeBO.py', 'training.train_GP_classifier', 'get_thompson_sample')] == 'reference'
    assert imp[('pareto.py', 'dataset.protein', 'ProteinPredictorDataset')] == 'dropin'
    assert imp[('pareto.py', 'models.pretraining.collaters', 'collate_fn_mapping')] == 'dropin'
    tg = {t: o for _, t, o in rep['targets']}
    for t in ('models.mdlm.MDLM', 'models.d3pm.D3PM', 'sampling.cls_guidance.Classifier_Guidance_Inpaint', 'sampling.daps.DAPS',
              'sampling.NOS.NOS', 'sampling.uncond.Uncond', 'problem.protein.ProteinPredictor',
              'dataset.protein.ProteinPredictorDataset', 'training.train_classifier.train_classifier'):
        assert tg[t] == 'dropin', (t, tg.get(t))
    for t in ('models.continuous.ContinuousModel', 'models.causalLM.CausalLM', 'sampling.cls_guidance.Classifier_Guidance_Continuous',
              'sampling.NOS.NOS_C', 'sampling.daps.DAPS_continuous', 'sampling.DPO.DPOInpaint'):
        assert tg[t] == 'reference', (t, tg.get(t))
    # targets that do not exist in the reference itself (defect D5 and leftovers) are reported, not resolved
    assert any('models.MDLM' in t for _, t, _ in rep['skipped_targets'])
    print(json.dumps(rep, indent=1))


def test_fallthrough_path_follows_sys_path(tmp_path, monkeypatch):
    """The fall-through list is recomputed when sys.path changes and never lists the drop-in directory twice."""
    from sgpo_amd.dropin._fallthrough import FallthroughPath
    own = tmp_path / 'dropin' / 'pkg'
    other = tmp_path / 'ref' / 'pkg'
    own.mkdir(parents=True)
    other.mkdir(parents=True)
    p = FallthroughPath([str(own)], 'pkg')
    assert list(p) == [str(own)]
    monkeypatch.syspath_prepend(str(tmp_path / 'ref'))
    assert list(p) == [str(own), str(other)] and p.fallthrough_dirs() == [str(other)]
    monkeypatch.syspath_prepend(str(tmp_path / 'dropin'))
    assert list(p) == [str(own), str(other)]
