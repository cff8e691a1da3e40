"""Checkpoints.

``load_model`` / ``load_official_model`` / ``save_model`` keep the reference's file format and behaviour
(src/utils/model.py:5-71): a checkpoint is ``{'epoch': int, 'state_dict': {...}}`` with canonical OIHW fp32 tensors
under the keys ``base.features.N.*`` / ``base.convdet.*``; ``module.`` prefixes of DataParallel checkpoints are
stripped, shape-mismatched / missing / unknown parameters are reported and skipped (never fatal), torchvision
SqueezeNet weights map by ``'base.' + key``.  Files written here load in the reference and vice versa.

``save_checkpoint`` / ``load_checkpoint`` add what the reference lacks for restarting a (multi-GPU) run exactly:
optimizer state (momentum buffers), LR-scheduler state and the RNG streams, under extra top-level keys that the
reference's ``load_model`` simply ignores.  One process per GPU: rank 0 writes, every rank reads.
"""
from __future__ import annotations

import os

import torch


def _strip_module_prefix(sd):
    out = {}
    for k, v in sd.items():
        if k.startswith('module') and not k.startswith('module_list'):
            out[k[7:]] = v
        else:
            out[k] = v
    return out


def _reconcile(model, state_dict, verbose=True):
    """The reference's tolerant matching (model.py:17-37).  Returns (state_dict to load, fully_loaded flag)."""
    model_sd = model.state_dict()
    ok = True
    for layer in list(state_dict):
        if layer in model_sd:
            if state_dict[layer].shape != model_sd[layer].shape:
                ok = False
                if verbose:
                    print('Skip loading param {}, required shape{}, loaded shape{}.'.format(
                        layer, model_sd[layer].shape, state_dict[layer].shape))
                state_dict[layer] = model_sd[layer]
        else:
            ok = False
            if verbose:
                print('Drop param {} in pre-trained model.'.format(layer))
    for layer in model_sd:
        if layer not in state_dict:
            ok = False
            if verbose:
                print('Param {} not found in pre-trained model.'.format(layer))
            state_dict[layer] = model_sd[layer]
    return state_dict, ok


def load_model(model, model_path, verbose=True):
    checkpoint = torch.load(model_path, map_location='cpu', weights_only=False)
    if verbose:
        print('loaded model {}, epoch {}'.format(model_path, checkpoint['epoch']))
    state_dict, ok = _reconcile(model, _strip_module_prefix(checkpoint['state_dict']), verbose)
    model.load_state_dict(state_dict, strict=False)
    if verbose:
        print('Model successfully loaded.' if ok else 'The model does not fully load the pre-trained weight.')
    return model


def load_official_model(model, model_path, verbose=True):
    """torchvision SqueezeNet state_dict (``features.N...``) -> ``base.features.N...`` (model.py:42-62); writes the
    converted checkpoint next to the original like the reference does."""
    state_dict = torch.load(model_path, map_location='cpu', weights_only=False)
    state_dict = {'base.' + k: v for k, v in state_dict.items()}
    converted = model_path.replace('.pth', '_converted.pth')
    torch.save({'epoch': 0, 'state_dict': state_dict}, converted)
    return load_model(model, converted, verbose)


def _unwrap(model):
    return model.module if hasattr(model, 'module') and isinstance(getattr(model, 'module'), torch.nn.Module) else model


def save_model(model, path, epoch):
    torch.save({'epoch': epoch, 'state_dict': _unwrap(model).state_dict()}, path)


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def _is_rank0():
    d = _dist()
    return d is None or d.get_rank() == 0


def _model_device(model):
    """The GPU this process computes on: the device of the model's parameters (a process that set ``cfg.device='cuda:N'`` without
    ``torch.cuda.set_device`` has another *current* device), else the current device, else None."""
    for p in _unwrap(model).parameters():
        if p.is_cuda:
            return p.device
        break
    return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None


def _rng_state_here(model):
    """This process's random streams: the CPU generator and the generator of the GPU the model lives on."""
    st = {'torch': torch.get_rng_state()}
    dev = _model_device(model)
    if dev is not None:
        st['cuda'] = torch.cuda.get_rng_state(dev)
    base = getattr(_unwrap(model), 'base', None)
    if base is not None and hasattr(base, 'get_dropout_rng'):
        st['sqd_dropout'] = base.get_dropout_rng()          # (seed, step) of the counter-based dropout in front of ConvDet, or None
    return st


def save_checkpoint(path, model, optimizer=None, lr_scheduler=None, epoch=0, rng=True, extra=None, all_ranks=True):
    """Reference-compatible file plus optimizer / scheduler / RNG state.  Tensors are moved to the CPU; written
    atomically (tmp + rename) by rank 0.

    Collective contract (one process per GPU): with ``all_ranks=True`` (default) EVERY rank must call this -- the random
    streams differ per rank (dropout masks), so each rank's state is gathered into the file, and the call ends with a
    barrier: when it returns on any rank the file is complete and ``load_checkpoint`` may read it.  A caller that keeps
    the reference's ``if rank == 0: save(...)`` pattern (src/train.py:70-78 saves from its single process) must pass
    ``all_ranks=False``: no collective is entered, only the calling rank's streams are stored and ranks > 0 resume from
    stream 0 with a warning (not an exact multi-GPU resume)."""
    d = _dist()
    multi = d is not None and d.get_world_size() > 1
    rng_states = None
    if rng:
        mine = _rng_state_here(model)
        if multi and all_ranks:
            rng_states = [None] * d.get_world_size() if d.get_rank() == 0 else None
            d.gather_object(mine, rng_states, dst=0)
        else:
            rng_states = [mine]
    if multi and all_ranks:
        try:
            if _is_rank0():
                _write_checkpoint(path, model, optimizer, lr_scheduler, epoch, rng, rng_states, extra)
        finally:
            d.barrier()                       # the file is in place before any rank goes on (and may load it)
        return
    if multi and not all_ranks and not _is_rank0():
        import warnings
        warnings.warn('save_checkpoint(all_ranks=False) called on a rank > 0: nothing written (rank 0 writes the file)')
        return
    _write_checkpoint(path, model, optimizer, lr_scheduler, epoch, rng, rng_states, extra)


def _write_checkpoint(path, model, optimizer, lr_scheduler, epoch, rng, rng_states, extra):
    data = {'epoch': epoch,
            'state_dict': {k: v.detach().cpu() for k, v in _unwrap(model).state_dict().items()}}
    if optimizer is not None:
        data['optimizer'] = optimizer.state_dict()
    if lr_scheduler is not None:
        data['lr_scheduler'] = lr_scheduler.state_dict()
    if rng:
        data['rng'] = {'per_rank': rng_states}
    if extra:
        data['extra'] = extra
    tmp = path + '.tmp'
    torch.save(data, tmp)
    os.replace(tmp, path)


def load_checkpoint(path, model, optimizer=None, lr_scheduler=None, restore_rng=True, verbose=False):
    """Inverse of ``save_checkpoint`` (every rank reads the file and takes ITS random streams back); also accepts plain
    reference checkpoints (then only the weights and the epoch come back).  Returns the epoch stored in the file."""
    import warnings
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    target = _unwrap(model)
    state_dict, _ = _reconcile(target, _strip_module_prefix(ckpt['state_dict']), verbose)
    target.load_state_dict(state_dict, strict=False)
    if optimizer is not None and 'optimizer' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer'])            # torch casts the state to each parameter's device
    if lr_scheduler is not None and 'lr_scheduler' in ckpt:
        lr_scheduler.load_state_dict(ckpt['lr_scheduler'])
    if restore_rng and 'rng' in ckpt:
        d = _dist()
        rank = d.get_rank() if d is not None else 0
        world = d.get_world_size() if d is not None else 1
        states = ckpt['rng'].get('per_rank')
        if states is None:                                      # files written before the per-rank format
            cuda = ckpt['rng'].get('cuda') or []
            states = [{'torch': ckpt['rng']['torch'], **({'cuda': cuda[0]} if len(cuda) else {})}]
        if len(states) != world:
            warnings.warn(f'checkpoint holds the random streams of {len(states)} rank(s), this job has {world}: '
                          f'rank {rank} resumes from stream {rank % len(states)} (not an exact resume)')
        st = states[rank % len(states)]
        torch.set_rng_state(st['torch'])
        dev = _model_device(model)
        if dev is not None and 'cuda' in st:
            torch.cuda.set_rng_state(st['cuda'], dev)
        base = getattr(_unwrap(model), 'base', None)
        if dev is not None and base is not None and st.get('sqd_dropout') is not None and hasattr(base, 'set_dropout_rng'):
            seed, step = st['sqd_dropout']
            # a file with fewer streams than ranks: every rank continues the saved step on a seed of its own
            base.set_dropout_rng(seed + (rank // len(states)) * 0x9e3779b97f4a7c15 if len(states) != world else seed, step, dev)
        elif base is not None and getattr(base, 'dropout', None) is not None and st.get('sqd_dropout') is None:
            warnings.warn('checkpoint carries no dropout stream (written before the counter-based dropout): the dropout in front of '
                          'ConvDet restarts at step 0 of the stream derived from the restored torch seed (masks of the first run repeat)')
        from .trainer import mark_rank_streams_set
        # all ranks' own streams are back: a later attach_data_parallel / Trainer(...) must not re-seed them.  Fewer streams than
        # ranks: several ranks now share a torch stream, so the next attach has to offset them again
        mark_rank_streams_set(len(states) == world)
    return ckpt['epoch']
