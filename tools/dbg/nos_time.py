"""NOS step at config-4 size (B = 200, L = 389, real DiT dims): kernel times from the library's per-kind event
profile plus a checksum of (delta, probs, xs) so that two builds (SGPO_HIP_LIB) can be compared bit for bit.
    python tools/dbg/nos_time.py [masked_fraction ...]
    NOS_SAVE=/tmp/a.pt: also save the outputs of the last fraction; NOS_CMP=/tmp/a.pt: compare against a saved file."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgpo_amd import runtime as rt, weights as W, _lib as L


def main():
    fracs = [float(v) for v in sys.argv[1:]] or [1.0, 0.5, 0.1]
    sd = W.dit_synthetic_state(seed=1, vocab=31, hidden=512, cond_dim=128, n_blocks=12, n_heads=8)
    net = rt.DitDenoiser(sd)
    B, Lq, steps = 200, 389, 5
    g = torch.Generator().manual_seed(5)
    w_cls = torch.randn(512, generator=g) * 0.05
    for f in fracs:
        x = torch.randint(0, 28, (B, Lq), generator=g)
        x[torch.rand(B, Lq, generator=g) < f] = 28
        u = torch.rand(B, Lq, 31, generator=g)
        hid = torch.randn(B, Lq, 512, generator=g)
        args = (net, x.cuda(), w_cls.cuda(), steps, 0.5, 0.1, 0.6, 0.55, 28)
        kw = dict(want_delta=True, noise=u.cuda(), hidden_in=hid.cuda())
        rt.nos_denoise(*args, **kw)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            xs, p, d = rt.nos_denoise(*args, **kw)
        ev[1].record()
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for t in (xs, p, d):
            h.update(t.cpu().numpy().tobytes())
        import ctypes as C
        lib = L.load()
        lib.sgpo_prof_enable(1, 64)
        rt.nos_denoise(*args, **kw)
        torch.cuda.synchronize()
        n = L.PROF_KINDS
        ms, cnt, work = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)()
        lib.sgpo_prof_read(n, ms, cnt, work)
        lib.sgpo_prof_enable(0, 0)
        print('   ' + ', '.join(f'{lib.sgpo_prof_kind_name(k).decode()} {ms[k] * 1e3:.0f} us / {cnt[k]}' for k in range(n) if cnt[k]))
        print(f'masked {f:.2f}: {ev[0].elapsed_time(ev[1]) / 5 * 1e3:8.1f} us per nos_denoise ({steps} inner steps, hidden_in given)  sha {h.hexdigest()[:16]}', flush=True)


    if os.environ.get('NOS_SAVE'):
        torch.save({'xs': xs.cpu(), 'p': p.cpu(), 'd': d.cpu()}, os.environ['NOS_SAVE'])
    if os.environ.get('NOS_CMP'):
        o = torch.load(os.environ['NOS_CMP'])
        for k, t in (('xs', xs), ('p', p), ('d', d)):
            t = t.cpu()
            ne = (t != o[k])
            print(f'{k}: {int(ne.sum())} of {t.numel()} elements differ; max |diff| {float((t.double() - o[k].double()).abs().max()):.3e}; max |ref| {float(o[k].double().abs().max()):.3e}')


if __name__ == '__main__':
    main()
