"""Checkpoint compatibility and exact resume (reference: src/utils/model.py:5-71)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import checkpoint as ck
from squeezedet_pytorch_amd import synthetic


class _Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.base = nn.Sequential(nn.Conv2d(3, 4, 3), nn.ReLU(), nn.Conv2d(4, 2, 1))

    def forward(self, x):
        return self.base(x)


def test_reference_format_roundtrip_and_tolerant_loading(tmp_path, capsys):
    torch.manual_seed(0)
    m = _Tiny()
    p = str(tmp_path / "m.pth")
    ck.save_model(m, p, 7)
    raw = torch.load(p, weights_only=False)
    assert set(raw) == {"epoch", "state_dict"} and raw["epoch"] == 7                 # the reference's file format
    assert list(raw["state_dict"]) == list(m.state_dict())
    m2 = ck.load_model(_Tiny(), p)
    for a, b in zip(m.state_dict().values(), m2.state_dict().values()):
        assert torch.equal(a, b)
    assert "Model successfully loaded." in capsys.readouterr().out
    # DataParallel-style 'module.' prefix, one wrong shape, one unknown and one missing parameter
    sd = {"module." + k: v.clone() for k, v in m.state_dict().items()}
    sd["module.base.0.weight"] = torch.zeros(5, 3, 3, 3)
    sd["module.extra.weight"] = torch.zeros(1)
    del sd["module.base.2.bias"]
    torch.save({"epoch": 1, "state_dict": sd}, p)
    fresh = _Tiny()
    before = {k: v.clone() for k, v in fresh.state_dict().items()}
    ck.load_model(fresh, p)
    out = capsys.readouterr().out
    assert "Skip loading param base.0.weight" in out and "Drop param extra.weight" in out
    assert "Param base.2.bias not found" in out and "does not fully load" in out
    assert torch.equal(fresh.state_dict()["base.0.weight"], before["base.0.weight"])   # kept its own value
    assert torch.equal(fresh.state_dict()["base.0.bias"], m.state_dict()["base.0.bias"])


def test_official_squeezenet_key_mapping(tmp_path):
    m = _Tiny()
    tv = {k[len("base."):]: v.clone() + 1 for k, v in m.state_dict().items()}        # torchvision-style keys
    p = str(tmp_path / "squeezenet1_1-f364aa15.pth")
    torch.save(tv, p)
    ck.load_official_model(m, p, verbose=False)
    assert os.path.exists(p.replace(".pth", "_converted.pth"))
    for k, v in m.state_dict().items():
        assert torch.equal(v, tv[k[len("base."):]])


def test_squeezedet_state_dict_contract_on_cpu(tmp_path):
    """64 tensors, canonical OIHW fp32, the reference's key names -- file written here has them all."""
    from squeezedet_pytorch_amd.model import SqueezeDetWithLoss
    cfg = sqd.make_cfg(arch="squeezedet", device="cpu")
    m = SqueezeDetWithLoss(cfg)
    m.load_state_dict(synthetic.make_state_dict("squeezedet", seed=3), strict=True)
    p = str(tmp_path / "sqd.pth")
    ck.save_model(m, p, 280)
    sd = torch.load(p, weights_only=False)["state_dict"]
    assert len(sd) == 64 and all(v.dtype == torch.float32 for v in sd.values())
    assert tuple(sd["base.features.0.weight"].shape) == (64, 3, 3, 3)
    assert tuple(sd["base.convdet.weight"].shape) == (72, 768, 3, 3)
    assert tuple(sd["base.features.3.expand3x3.weight"].shape) == (64, 16, 3, 3)
    m2 = ck.load_model(SqueezeDetWithLoss(cfg), p, verbose=False)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_exact_resume_of_optimizer_scheduler_and_rng_cpu(tmp_path):
    def make():
        torch.manual_seed(1)
        m = _Tiny()
        opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        return m, opt, torch.optim.lr_scheduler.StepLR(opt, 2, gamma=0.5)

    def step(m, opt, sched):
        x = torch.randn(2, 3, 8, 8)                     # consumes the global RNG stream
        opt.zero_grad(); m(x).square().mean().backward(); opt.step(); sched.step()

    m, opt, sched = make()
    for _ in range(3):
        step(m, opt, sched)
    p = str(tmp_path / "resume.pth")
    ck.save_checkpoint(p, m, opt, sched, epoch=3)
    for _ in range(3):
        step(m, opt, sched)
    want = [v.clone() for v in m.state_dict().values()]
    m2, opt2, sched2 = make()
    assert ck.load_checkpoint(p, m2, opt2, sched2) == 3
    assert sched2.get_last_lr() == pytest.approx([0.005])
    for _ in range(3):
        step(m2, opt2, sched2)
    assert all(torch.equal(a, b) for a, b in zip(want, m2.state_dict().values()))
    # the extended file is still a valid reference checkpoint
    ck.load_model(_Tiny(), p, verbose=False)


@pytest.mark.gpu
def test_exact_resume_squeezedet_training_gpu(tmp_path):
    from squeezedet_pytorch_amd.model import SqueezeDetWithLoss
    size = (64, 96)

    def make():
        cfg = sqd.make_cfg(arch="squeezedet", input_size=size, dropout_prob=0.5)
        m = SqueezeDetWithLoss(cfg)
        m.load_state_dict(synthetic.make_state_dict("squeezedet", seed=1234), strict=True)
        m = m.cuda().train()
        opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        return cfg, m, opt, torch.optim.lr_scheduler.StepLR(opt, 2, gamma=0.5)

    def step(cfg, m, opt, sched, it):
        x = synthetic.make_images(2, size, seed=10 + it).cuda()
        gt = synthetic.make_gt(2, cfg.anchors, size, seed=30 + it, min_boxes=2, max_boxes=3).cuda()
        loss, _ = m({"image": x, "gt": gt})               # dropout draws from the CUDA RNG stream
        opt.zero_grad(); loss.mean().backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0); opt.step(); sched.step()
        return float(loss.mean())

    torch.manual_seed(5); torch.cuda.manual_seed_all(5)
    cfg, m, opt, sched = make()
    for it in range(2):
        step(cfg, m, opt, sched, it)
    p = str(tmp_path / "sqd_resume.pth")
    ck.save_checkpoint(p, m, opt, sched, epoch=2)
    tail = [step(cfg, m, opt, sched, it) for it in range(2, 4)]
    want = [v.clone() for v in m.state_dict().values()]
    cfg2, m2, opt2, sched2 = make()
    assert ck.load_checkpoint(p, m2, opt2, sched2) == 2
    tail2 = [step(cfg2, m2, opt2, sched2, it) for it in range(2, 4)]
    assert tail == tail2                                                         # bitwise: same losses ...
    assert all(torch.equal(a, b) for a, b in zip(want, m2.state_dict().values()))   # ... and same weights


def _resume_worker(rank, world, port, path, out_dir):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from squeezedet_pytorch_amd import checkpoint as ckp
    torch.manual_seed(100 + rank)                                   # every rank its own dropout stream
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Dropout(0.5))

    class Wrapped(torch.nn.Module):                                  # a `.module` wrapper like (Distributed)DataParallel
        def __init__(self, m):
            super().__init__()
            self.module = m
    torch.rand(3)                                                    # advance the stream a little
    ckp.save_checkpoint(path, Wrapped(net), epoch=3)                 # collective: every rank calls it
    want = torch.rand(8)                                             # what an uninterrupted run draws next
    torch.manual_seed(999)                                           # "restart": streams are somewhere else entirely
    net2 = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Dropout(0.5))
    dist.barrier()
    assert ckp.load_checkpoint(path, Wrapped(net2)) == 3
    got = torch.rand(8)
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values())) or rank != 0
    torch.save({'want': want, 'got': got}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_resume_restores_each_ranks_random_stream(tmp_path):
    """Multi-process resume: save_checkpoint gathers every rank's RNG state, load_checkpoint hands each rank ITS stream
    back (rank 1 must not continue with rank 0's dropout masks); weights load into ``.module`` wrappers."""
    import torch.multiprocessing as mp
    world, port = 2, 33500 + (os.getpid() % 2000)
    path = str(tmp_path / 'dp_resume.pth')
    mp.spawn(_resume_worker, args=(world, port, path, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(str(tmp_path / 'r0.pt')); r1 = torch.load(str(tmp_path / 'r1.pt'))
    assert torch.equal(r0['want'], r0['got']) and torch.equal(r1['want'], r1['got'])
    assert not torch.equal(r0['got'], r1['got'])
    ckpt = torch.load(path, weights_only=False)
    assert len(ckpt['rng']['per_rank']) == 2 and set(ckpt['state_dict']) == {'0.weight', '0.bias'}


def _resume_then_trainer_worker(rank, world, port, path, out_dir):
    """The reference's resume order (src/train.py: load the checkpoint, THEN build Trainer, whose set_device attaches the
    data-parallel exchange): the attach must not re-seed the streams load_checkpoint has just restored."""
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import checkpoint as ckp, trainer as tr
    from squeezedet_pytorch_amd.model import SqueezeDetWithLoss
    cfg = sqd.make_cfg(input_size=(64, 96), device='cpu', gpus=[0, 1], chunk_sizes=[1, 1])

    def build():
        m = SqueezeDetWithLoss(cfg)
        opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
        return m, opt, torch.optim.lr_scheduler.StepLR(opt, 60, 0.5)
    torch.manual_seed(7)
    m, opt, sched = build()
    t = tr.Trainer(m, opt, sched, cfg)                              # first attach: ranks > 0 get their own stream (once)
    first = torch.initial_seed()
    tr.attach_data_parallel(m, opt)                                 # INTEGRATION.md recipe + Trainer = two attaches: no second offset
    assert torch.initial_seed() == first and first == 7 + rank
    torch.rand(5)
    ckp.save_checkpoint(path, m, opt, sched, epoch=1)               # collective, ends with a barrier: the file is complete here
    want = torch.rand(8)                                            # the uninterrupted run's next draw
    # "restart" in the same processes: fresh objects, streams elsewhere, and the process-level flag cleared as in a new process
    torch.manual_seed(999)
    tr.mark_rank_streams_set(False)
    m2, opt2, sched2 = build()
    assert ckp.load_checkpoint(path, m2, opt2, sched2) == 1
    tr.Trainer(m2, opt2, sched2, cfg)                               # set_device -> attach_data_parallel: must leave the streams alone
    got = torch.rand(8)
    torch.save({'want': want, 'got': got}, os.path.join(out_dir, f't{rank}.pt'))
    assert t.exchange is not None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_load_checkpoint_then_trainer_keeps_restored_streams(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, 35500 + (os.getpid() % 2000)
    path = str(tmp_path / 'dp_resume_trainer.pth')
    mp.spawn(_resume_then_trainer_worker, args=(world, port, path, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(str(tmp_path / 't0.pt')); r1 = torch.load(str(tmp_path / 't1.pt'))
    assert torch.equal(r0['want'], r0['got']) and torch.equal(r1['want'], r1['got'])
    assert not torch.equal(r0['got'], r1['got'])


def _rank0_only_save_worker(rank, world, port, path, out_dir):
    import sys
    import warnings
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from squeezedet_pytorch_amd import checkpoint as ckp
    net = torch.nn.Linear(3, 3)
    if rank == 0:                                                    # the reference's pattern: only rank 0 saves -- must not hang
        ckp.save_checkpoint(path, net, epoch=5, all_ranks=False)
    dist.barrier()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        assert ckp.load_checkpoint(path, torch.nn.Linear(3, 3)) == 5
    assert any('not an exact resume' in str(x.message) for x in w)  # one stream in the file, two ranks
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_rank0_only_save_does_not_deadlock(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, 37500 + (os.getpid() % 2000)
    mp.spawn(_rank0_only_save_worker, args=(world, port, str(tmp_path / 'r0only.pth'), str(tmp_path)), nprocs=world, join=True)
