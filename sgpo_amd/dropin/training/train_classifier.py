"""Drop-in `training.train_classifier` (reference: training/train_classifier.py:7-64; SURVEY.md section 8
row f-3): the same `train_classifier(classifier, model, dataloader, train_config, ...)` call, with the
optimiser steps on the GPU (`sgpo_mlp_trainer_*`: forward, backward and Adam fused into four launches per step).

`train_classifier_ensemble` advances all ensemble members in lockstep in ONE process -- the reference spawns one
process per member (iterativeBO.py:157-173) and reloads the pickled models (iterativeBO.py:176-179).

Randomness: the reference draws the timesteps from torch's global CPU generator, the noise of `model.q_sample`
from the device generator and the batch order from the DataLoader's sampler; here `generator` (or a generator
seeded from torch's global seed) drives all three, per member.  Same distributions, not the same streams.
"""
import os

import torch

from models.base import cfg_get


def _collect(dataloader, device):
    """The whole data set as tensors: x (n, L) int64, y (n,) float32 (one pass over the loader's batches)."""
    xs, ys = [], []
    for x, y in dataloader:
        xs.append(torch.as_tensor(x).to(device, torch.int64))
        ys.append(torch.as_tensor(y).to(device, torch.float32).reshape(-1))
    return torch.cat(xs), torch.cat(ys)


def train_classifier_ensemble(classifiers, model, dataloader, train_config, save_dir=None, project_fn=None,
                              time_conditioned=True, generator=None):
    """Trains every `ProteinPredictor` of `classifiers` (same architecture) on the loader's data set, each with its
    own batch order, timesteps and noise: n_epochs passes (train_classifier.py:30-42) of batches of
    `dataloader.batch_size` (the last one ragged, as DataLoader's drop_last=False), Adam(lr) on MSELoss.
    Returns the classifiers with the trained weights loaded (and saved like the reference when save_dir is set)."""
    from sgpo_amd import runtime as rt
    device = rt.init(getattr(model, '_dev', None) or getattr(model, 'device', None))
    E = len(classifiers)
    m0 = classifiers[0].model
    states = [{k: v.detach() for k, v in c.model.state_dict().items()} for c in classifiers]
    trainer = rt.MlpEnsembleTrainer(states, m0.read_positions(), alphabet=m0.alphabet_size, hidden=m0.hidden_dim,
                                    time_conditioned=m0.time_conditioned, device=device)
    X, Y = _collect(dataloader, device)
    n, Lq = X.shape
    bs = int(getattr(dataloader, 'batch_size', None) or cfg_get(train_config, 'batch_size'))
    if bs > 64:
        raise ValueError(f'the fused trainer takes batches of at most 64 sequences (got batch_size {bs})')
    lr = float(cfg_get(train_config, 'learning_rate'))
    n_epochs = cfg_get(train_config, 'n_epochs') if time_conditioned else cfg_get(train_config, 'n_epochs_no_time')
    if generator is None:
        generator = torch.Generator(device=device)
        generator.manual_seed(int(torch.initial_seed()) & 0x7FFFFFFF)
    T = int(model.timestep) if time_conditioned else 1
    # Batches are prepared a group at a time (gather, timesteps, q_sample, project_fn over up to 64 batches per member
    # in one call each -- and, when an epoch is shorter than that, over several epochs at once), so that a step costs a
    # slice and the five training launches instead of a dozen small torch kernels.  Every sequence still gets its own
    # timestep and noise, every member its own order in every epoch.
    group = 64 * bs

    def run_group(idx):                                               # idx: (E, G) rows of the data set, in batch order
        G = idx.shape[1]
        x = X[idx]                                                    # (E, G, L)
        y = Y[idx]
        if time_conditioned:
            t = torch.randint(0, T, (E, G), device=device, generator=generator)           # train_classifier.py:36
            xt = model.q_sample(x.reshape(E * G, Lq), t.reshape(E * G)).reshape(E, G, Lq)       # :13
        else:
            t = torch.zeros(E, G, dtype=torch.int64, device=device)                       # :38
            xt = x
        if project_fn is not None:
            xt = project_fn(xt)                                                           # :14
        return xt, t.float(), y

    def run_batches(xt, tf, y):                                       # one epoch's (or one group's) rows of every member
        G = xt.shape[1]
        nb = G // bs                                                  # full batches, laid out (batch, member, row)
        if nb:
            xb = xt[:, :nb * bs].reshape(E, nb, bs, Lq).transpose(0, 1).contiguous()
            tb = tf[:, :nb * bs].reshape(E, nb, bs).transpose(0, 1).contiguous()
            yb = y[:, :nb * bs].reshape(E, nb, bs).transpose(0, 1).contiguous()
            trainer.steps(xb, tb, yb, lr=lr)                          # the nb full batches, one library call
        if G > nb * bs:                                               # the ragged last batch (drop_last=False)
            trainer.step(xt[:, nb * bs:].contiguous(), tf[:, nb * bs:].contiguous(), y[:, nb * bs:].contiguous(), lr=lr)

    def shuffles(k):                                                  # k epochs x E members: shuffle=True, one sort
        return torch.rand(k, E, n, device=device, generator=generator).argsort(dim=-1)

    done, n_epochs = 0, int(n_epochs)
    while done < n_epochs:
        if n <= group:
            k = max(1, min(n_epochs - done, group // n, 262144 // max(1, E * n)))
            perm = shuffles(k)                                        # (k, E, n)
            xt, tf, y = run_group(perm.permute(1, 0, 2).reshape(E, k * n))
            for j in range(k):                                        # epoch by epoch: the ragged batch ends each one
                run_batches(xt[:, j * n:(j + 1) * n], tf[:, j * n:(j + 1) * n], y[:, j * n:(j + 1) * n])
        else:
            k = 1
            perm = shuffles(1)[0]
            for g0 in range(0, n, group):
                run_batches(*run_group(perm[:, g0:g0 + group]))
        done += k
    for c, sd in zip(classifiers, trainer.state_dicts()):
        c.model.load_state_dict(sd)
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
        for e, c in enumerate(classifiers):
            torch.save(c, os.path.join(save_dir, f'classifier_{e}.pt'))                   # train_classifier.py:62
    return classifiers


def train_classifier(classifier, model, dataloader, train_config, save_dir=None, project_fn=None, ensemble_idx=0,
                     time_conditioned=True):
    """The reference's signature: one member (one call per spawned process there)."""
    train_classifier_ensemble([classifier], model, dataloader, train_config, save_dir=None, project_fn=project_fn,
                              time_conditioned=time_conditioned)
    if save_dir is None:
        return classifier
    os.makedirs(save_dir, exist_ok=True)
    torch.save(classifier, os.path.join(save_dir, f'classifier_{ensemble_idx}.pt'))
    return classifier
