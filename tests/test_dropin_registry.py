"""the hydra `_target_` strings and the driver-side imports must resolve to the drop-in
package with the reference's call signatures (CPU check: import + signatures, no compute)."""
import importlib
import inspect

import sgpo_amd


def _get(path):
    mod, name = path.rsplit('.', 1)
    return getattr(importlib.import_module(mod), name)


def test_registry_targets_resolve_with_reference_signatures():
    sgpo_amd.install_dropin()
    want = {
        'models.mdlm.MDLM': ['model_name', 'config', 'tokenizer', 'num_steps', 'seq_len', 'device'],          # models/mdlm.py:23
        'models.d3pm.D3PM': ['model_name', 'seq_len', 'device'],                                              # models/d3pm.py:10
        'sampling.cls_guidance.Classifier_Guidance_Inpaint': ['net', 'data_config', 'forward_op', 'temperature', 'n_max_mutations', 'device'],
        'sampling.daps.DAPS': ['net', 'forward_op', 'num_steps', 'ode_steps', 'mh_steps', 'alpha', 'beta', 'max_dist'],
        'sampling.NOS.NOS': ['net', 'data_config'],
        'sampling.uncond.Uncond': ['net', 'forward_op'],
        'problem.protein.ProteinPredictor': ['data_config', 'model_config', 'checkpoint', 'device'],          # problem/protein.py:14
        'dataset.protein.ProteinPredictorDataset': ['data_config', 'tokenizer', 'n_random_init', 'n_max_mutations', 'from_fasta'],
    }
    for path, params in want.items():
        cls = _get(path)
        sig = list(inspect.signature(cls.__init__).parameters)[1:]
        for p in params:
            assert p in sig, (path, p, sig)
    for path, params in {
        'training.train_classifier.train_classifier': ['classifier', 'model', 'dataloader', 'train_config', 'save_dir', 'project_fn',
                                                       'ensemble_idx', 'time_conditioned'],                   # train_classifier.py:21
        'fitness.inference_oracle.inference_oracle': ['split', 'protein', 'model_path', 'impose_penalty'],    # inference_oracle.py:19
    }.items():
        assert list(inspect.signature(_get(path)).parameters)[:len(params)] == params, path
    for name in ('get_start', 'p_sample', 'q_sample', 'get_guided_rates', '_nos_denoise', 'sample'):
        assert callable(getattr(_get('models.mdlm.MDLM'), name)), name
    for name in ('get_start', 'p_sample', 'q_sample', 'get_guided_rates'):
        assert callable(getattr(_get('models.d3pm.D3PM'), name)), name
    for algo in ('sampling.cls_guidance.Classifier_Guidance_Inpaint', 'sampling.daps.DAPS', 'sampling.NOS.NOS'):
        for name in ('inference', 'update_model', 'project', 'project_sequences'):
            assert callable(getattr(_get(algo), name)), (algo, name)
    assert callable(_get('dataset.protein.ProteinPredictorDataset').update_data)


def test_driver_side_imports():
    sgpo_amd.install_dropin()
    import torch
    from models.pretraining.collaters import collate_fn_mapping          # pareto.py:10, iterativeBO.py:13
    from sgpo_amd.tokenizer import Tokenizer
    x, y = collate_fn_mapping['cls_guidance'](tokenizer=Tokenizer())([(('ACDE',), 1.0), (('WY#C',), 2.0)])
    assert x.tolist() == [[0, 1, 2, 3], [18, 19, 28, 1]] and y.tolist() == [1.0, 2.0] and x.dtype == torch.int64
    from problem.protein import MLPModel, ProteinPredictor               # iterativeBO.py:178 unpickles these
    for attr in ('embedder', 'time_embedder', 'mlp', 'cls_head'):
        assert attr in inspect.getsource(MLPModel.__init__)
    assert issubclass(ProteinPredictor, torch.nn.Module)


class _Hyper:        # a pickled object, as Lightning checkpoints with hyper-parameters carry
    pass


def test_checkpoint_importer_uses_the_non_executing_loader(tmp_path, monkeypatch):
    import pytest
    import torch
    from sgpo_amd import weights as W
    p = str(tmp_path / 'best_model.ckpt')
    torch.save({'state_dict': {'backbone.blocks.0.w': torch.ones(3), 'noise.sigma': torch.zeros(2)}}, p)
    sd = W.load_lightning_state(p, 'backbone.')
    assert list(sd) == ['blocks.0.w'] and sd['blocks.0.w'].dtype == torch.float32
    torch.save({'state_dict': {'backbone.w': torch.ones(3)}, 'hyper_parameters': _Hyper()}, p)
    monkeypatch.setenv('SGPO_TRUST_CHECKPOINT', '1')        # the former opt-in: no environment switch unpickles any more
    with pytest.raises(RuntimeError, match='non-executing loader refused'):
        W.load_lightning_state(p, 'backbone.')
    assert 'weights_only=False' not in inspect.getsource(W)
    with pytest.raises(KeyError):
        torch.save({'state_dict': {'other.w': torch.ones(3)}}, p)
        W.load_lightning_state(p, 'backbone.')
